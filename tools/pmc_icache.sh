export TMPDIR=/tmp
out=gpurun_out/pmc_icache
mkdir -p $out
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 tools/run_frames.py c3 2 single > "$out/$name.log" 2>&1 || { echo "FAILED $name"; return 1; }; }
run ic1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES && \
run ic2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $out render_kernel

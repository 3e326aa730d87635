# quick PMC: issue slots, TA busy, L2 hit for one library on C5 fused.  usage: pmc_quick.sh lib tag [kernel's ACCEL: 3 (80-byte nodes) or 2]
export TMPDIR=/tmp
export RT_LIB_PATH=$PWD/raytracer_amd/lib/$1
out=gpurun_out/pmcq_$2; mkdir -p $out
run() { name=$1; shift; timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --workload c5 --pipeline pooled --steps 2 --warmup 1 --no-cpu-baseline --no-verify > "$out/$name.log" 2>&1 || { echo "FAILED $name"; return 1; }; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM && \
run sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA && \
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && \
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE || exit 1
python3 tools/pmc_summary.py $out "pool_kernel<${3:-3}, false, true" 1 | grep "^#"

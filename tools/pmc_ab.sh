export TMPDIR=/tmp
for v in new HEAD; do
  if [ $v = HEAD ]; then export RT_LIB_PATH=$PWD/raytracer_amd/lib/libraytracer_amd_HEAD.so; else unset RT_LIB_PATH; fi
  out=gpurun_out/r02j/pmc_$v; mkdir -p $out
  run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --pipeline pooled > "$out/$name.log" 2>&1 || { echo "FAILED $name"; return 1; }; }
  run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM && \
  run sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA && \
  run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE || exit 1
  python3 tools/pmc_summary.py $out "pool_kernel<2, false, false" 1 > gpurun_out/r02j/pmc_$v.txt
done
paste -d'|' gpurun_out/r02j/pmc_new.txt gpurun_out/r02j/pmc_HEAD.txt | cut -c1-230

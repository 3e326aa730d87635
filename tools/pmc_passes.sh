#!/bin/bash
# PMC counters of the benchmark's render kernel, one rocprofv3 pass per counter group
# (never combined with tracing).  Usage: [BENCH_ARGS="--workload c5"] tools/pmc_passes.sh [outdir]
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
run() { name=$1; shift; timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify $BENCH_ARGS > "$out/$name.log" 2>&1 || { echo "FAILED $name"; return 1; }; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM && \
run sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA && \
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && \
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE && \
run flops SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT && \
run fetch FETCH_SIZE && run write WRITE_SIZE

#!/bin/bash
# HBM-side bytes of the benchmark's render kernel: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes.
# Usage: [BENCH_ARGS="--workload c5"] tools/pmc_hbm.sh outdir "kernel name part"
export TMPDIR=/tmp
out=${1:-gpurun_out/pmc_hbm}; kern=${2:-pool_kernel<2, false, false}
mkdir -p "$out"
run() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify $BENCH_ARGS > "$out/$name.log" 2>&1 || { echo "FAILED $name"; return 1; }; }
run fetch FETCH_SIZE && run write WRITE_SIZE && run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
python3 tools/pmc_summary.py "$out" "$kern" 1

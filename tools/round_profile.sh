#!/bin/bash
# Evidence for one round: bench lines, rocprofv3 kernel stats of the bench command and PMC passes; everything lands
# under gpurun_out/$1 (then copy the summaries into profiles/).  Every step under its own time limit; the first
# failure ends the script.
tag=${1:-r03}
out=gpurun_out/$tag
export TMPDIR=/tmp
mkdir -p $out
timeout -k 10 300 python bench.py > $out/bench_c3.json 2> $out/bench_c3.err || exit 1
timeout -k 10 300 python bench.py --frames-per-launch 1 --no-cpu-baseline > $out/bench_c3_fpl1.json 2> $out/bench_c3_fpl1.err || exit 1
for wl in c2 c5; do timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --steps 5 > $out/bench_$wl.json 2> $out/bench_$wl.err || exit 1; done
timeout -k 10 300 python bench.py --workload c4 --no-cpu-baseline --steps 2 --warmup 1 > $out/bench_c4.json 2> $out/bench_c4.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_c3 --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify > $out/prof_c3.log 2>&1 || exit 1
cp $out/prof_c3/*/*kernel_stats.csv $out/c3_kernel_stats.csv
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $out/prof_c5 --output-format csv -- python3 bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $out/prof_c5.log 2>&1 || exit 1
cp $out/prof_c5/*/*kernel_stats.csv $out/c5_kernel_stats.csv
# counters per FRAME: one frame per launch; then the fused launch the bench times (32 frames)
BENCH_ARGS="--pipeline pooled --frames-per-launch 1" bash tools/pmc_passes.sh $out/pmc_c3_pooled || exit 1
python3 tools/pmc_summary.py $out/pmc_c3_pooled "pool_kernel<2, false, false" 1 > $out/c3_pooled_pmc.txt
BENCH_ARGS="--pipeline pooled --frames-per-launch 32" bash tools/pmc_passes.sh $out/pmc_c3_fused || exit 1
python3 tools/pmc_summary.py $out/pmc_c3_fused "pool_kernel<2, false, true" 1 > $out/c3_pooled_fused32_pmc.txt
BENCH_ARGS="--workload c5 --pipeline pooled --frames-per-launch 1" bash tools/pmc_passes.sh $out/pmc_c5_pooled || exit 1
python3 tools/pmc_summary.py $out/pmc_c5_pooled "pool_kernel<3, false, false" 1 > $out/c5_pooled_pmc.txt
BENCH_ARGS="--workload c5 --pipeline pooled --frames-per-launch 32" bash tools/pmc_passes.sh $out/pmc_c5_fused || exit 1
python3 tools/pmc_summary.py $out/pmc_c5_fused "pool_kernel<3, false, true" 1 > $out/c5_pooled_fused32_pmc.txt
cat $out/bench_c3.json; head -4 $out/c3_kernel_stats.csv | cut -c1-160

# The four bench lines of a round (C3 with the CPU baseline), into gpurun_out/$1.
out=gpurun_out/${1:-bench}; mkdir -p $out
timeout -k 10 300 python bench.py > $out/bench_c3.json 2> $out/bench_c3.err || exit 1
for wl in c2 c5; do timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --steps 5 > $out/bench_$wl.json 2> $out/bench_$wl.err || exit 1; done
timeout -k 10 300 python bench.py --workload c4 --no-cpu-baseline --steps 2 --warmup 1 > $out/bench_c4.json 2> $out/bench_c4.err || exit 1

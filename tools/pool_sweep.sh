for rm in 8 16 24 32; do for b in 16 64; do echo "refill_min=$rm burst=$b: $(RT_REFILL_MIN=$rm RT_POOL_BURST=$b timeout -k 5 20 python3 tools/run_frames.py ${1:-c3} 4 pooled | tail -1)" || exit 1; done; done
RT_REFILL_MIN=16 RT_POOL_BURST=64 timeout -k 5 40 python3 tools/run_frames.py ${1:-c3} 2 pooled stats | tail -3

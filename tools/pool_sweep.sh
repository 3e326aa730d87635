for jm in 48 64; do for rm in 8 16 32; do for b in 8 16 32; do echo "job_min=$jm refill_min=$rm burst=$b: $(RT_POOL_JOB_MIN=$jm RT_REFILL_MIN=$rm RT_POOL_BURST=$b timeout -k 5 20 python3 tools/run_frames.py ${1:-c3} 4 pooled | tail -1)" || exit 1; done; done; done
RT_POOL_BURST=16 timeout -k 5 40 python3 tools/run_frames.py c3 2 pooled stats | tail -2

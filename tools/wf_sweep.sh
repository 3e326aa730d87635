for pc in 8 4 2 1; do for rm in 8 16 32; do echo "search_per_cu=$pc refill_min=$rm"; RT_WF_SEARCH_PER_CU=$pc RT_REFILL_MIN=$rm timeout -k 10 60 python3 tools/run_frames.py c3 3 staged | tail -1 || exit 1; done; done
for r in 1024 2048 4096; do echo "shade_range=$r"; RT_WF_SHADE_RANGE=$r timeout -k 10 60 python3 tools/run_frames.py c3 3 staged | tail -1 || exit 1; done

f() { echo -n "[$*] "; env "$@" timeout -k 10 200 python tools/batch_time.py c5 16 4 | sed -n 2p | sed 's/c5 1920x1080 16 frames per launch: //'; }
s() { echo -n "[$*] "; env "$@" timeout -k 10 120 python tools/ab_time.py c5 8 pooled | grep median | sed 's/libraytracer_amd.so *//'; }
f RT_X=0
f RT_POOL_BURST=12
f RT_POOL_BURST=20
f RT_POOL_BURST=24
f RT_POOL_SHADER_WAVES=1
f RT_POOL_SHADER_WAVES=3
f RT_POOL_JOB_MIN=24
f RT_POOL_JOB_MIN=48
f RT_REFILL_MIN=8
f RT_REFILL_MIN=24
f RT_X=0
s RT_X=0
s RT_POOL_SLOTS=768
s RT_POOL_SLOTS=1024
s RT_POOL_SLOTS=1152
s RT_POOL_SLOTS=1344
s RT_POOL_BURST=12
s RT_POOL_BURST=20
s RT_X=0

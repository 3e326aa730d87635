# Kernel time of one rank's share of a frame under an N-way strip split, N = 1, 2, 4, 8: one frame per launch for both
# kernels, then 8 frames per launch (rt_render_frames) against one, per frame.
set -e
w=${1:-c3}
for n in 1 2 4 8; do for mode in pooled single; do
RT_STRIP=9,$n,0 timeout -k 10 120 python tools/ab_time.py $w 8 $mode
done; done
for n in 1 2 4 8; do echo "N=$n:"; RT_STRIP=9,$n,0 timeout -k 10 200 python tools/batch_time.py $w 8 5; done

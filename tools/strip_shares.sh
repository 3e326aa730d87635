# Kernel time of one rank's share of a frame under an N-way strip split, N = 1, 2, 4, 8, for both kernels.
set -e
w=${1:-c3}
for n in 1 2 4 8; do for mode in pooled single; do
RT_STRIP=9,$n,0 timeout -k 10 120 python tools/ab_time.py $w 8 $mode
done; done

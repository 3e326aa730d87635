set -e
L=raytracer_amd/lib
for rep in 1 2; do
for w in c3 c2 c1; do
for lib in libraytracer_amd_HEAD.so libraytracer_amd.so; do
RT_LIB_PATH=$PWD/$L/$lib timeout -k 10 120 python tools/ab_time.py $w 10 single
done; done; done

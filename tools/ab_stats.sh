# Stage statistics of one workload for two builds of the library, one after the other on the same device.
set -e
L=raytracer_amd/lib
for lib in libraytracer_amd_HEAD.so libraytracer_amd.so; do
echo "== $lib"
RT_LIB_PATH=$PWD/$L/$lib timeout -k 10 120 python tools/run_frames.py ${1:-c3} 3 pooled stats | tail -8
done

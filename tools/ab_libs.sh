# Several builds of the library timed on the same workloads, alternating:  tools/ab_libs.sh "lib1.so lib2.so ..." "c3 c5" [mode]
set -e
L=raytracer_amd/lib
for rep in 1 2; do for w in $2; do for lib in $1; do
RT_LIB_PATH=$PWD/$L/$lib timeout -k 10 120 python tools/ab_time.py $w 10 ${3:-pooled}
done; done; done

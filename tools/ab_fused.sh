# Several builds of the library timed on fused launches (tools/batch_time.py), alternating:  tools/ab_fused.sh "lib1.so lib2.so" "c3 c5" [frames]
L=raytracer_amd/lib
for rep in 1 2; do for w in $2; do for lib in $1; do
echo -n "$lib $w: "; RT_LIB_PATH=$PWD/$L/$lib timeout -k 10 200 python tools/batch_time.py $w ${3:-8} 5 | grep "frames per launch\|bit-identical" | tr '\n' ' '; echo
done; done; done

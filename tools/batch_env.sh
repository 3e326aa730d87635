# tools/batch_time.py (fused frames) under several values of an environment variable:  tools/batch_env.sh VAR "v1 v2" workload frames
set -e
var=$1; vals=$2; w=${3:-c3}; f=${4:-8}
for rep in 1 2; do for v in $vals; do echo -n "$var=$v  "; env $var=$v timeout -k 10 200 python tools/batch_time.py $w $f 4 | sed -n 2p; done; done

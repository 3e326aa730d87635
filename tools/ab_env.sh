# One workload timed under several values of an environment variable:  tools/ab_env.sh VAR "v1 v2 ..." workload [mode]
set -e
var=$1; vals=$2; w=${3:-c3}; mode=${4:-pooled}
for rep in 1 2; do for v in $vals; do
echo -n "$var=$v  "; env $var=$v timeout -k 10 120 python tools/ab_time.py $w 10 $mode
done; done

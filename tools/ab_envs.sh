# One workload timed under several environments (each a quoted "VAR=v VAR2=w" string; "-" = none), alternating, then the
# traversal counts of each:  tools/ab_envs.sh workload mode "env1" "env2" ...
w=$1; mode=$2; shift 2
for rep in 1 2; do for e in "$@"; do
  [ "$e" = "-" ] && e="RT_NONE=0"
  echo -n "[$e] "; env $e timeout -k 10 120 python tools/ab_time.py $w 10 $mode | grep median || exit 1
done; done
for e in "$@"; do
  [ "$e" = "-" ] && e="RT_NONE=0"
  echo -n "[$e] "; env $e timeout -k 10 120 python tools/run_frames.py $w 2 $mode stats | grep "^frame 1" || exit 1
done

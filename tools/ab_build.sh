#!/bin/bash
# Build the library from another git revision next to the current one, for A/B timing:
#   tools/ab_build.sh <rev>   ->  raytracer_amd/lib/libraytracer_amd_<rev>.so
set -e
rev=$1
root=$(cd "$(dirname "$0")/.." && pwd)
wt=/tmp/rt_ab_$rev
rm -rf "$wt"; git -C "$root" worktree prune
git -C "$root" worktree add --detach "$wt" "$rev" >/dev/null
make -C "$wt/raytracer_amd/csrc" -j8 >/dev/null
cp "$wt/raytracer_amd/lib/libraytracer_amd.so" "$root/raytracer_amd/lib/libraytracer_amd_$rev.so"
git -C "$root" worktree remove --force "$wt"
echo "built raytracer_amd/lib/libraytracer_amd_$rev.so"

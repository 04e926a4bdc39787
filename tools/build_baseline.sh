#!/bin/bash
# Build libmadsim_hip_<tag>.so from another git revision's kernel sources (A/B partner for tools/gpu_round.sh ab:...):
#   tools/build_baseline.sh <git-rev> <tag>
set -e
rev=$1; tag=$2; R=$(cd "$(dirname "$0")/.." && pwd); W=/tmp/madsim_ab_$tag
rm -rf "$W"; git -C "$R" worktree prune; git -C "$R" worktree add -f --detach "$W" "$rev" > /dev/null
make -C "$W/madsim_amd/csrc" -s
cp "$W/madsim_amd/libmadsim_hip.so" "$R/madsim_amd/libmadsim_hip_$tag.so"
git -C "$R" worktree remove --force "$W"
echo "built madsim_amd/libmadsim_hip_$tag.so from $rev"

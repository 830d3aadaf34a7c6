#!/bin/bash
# Build the HIP library of another git revision into build/lib_<name>.so (A/B timing on one GPU box: tools/ab.sh).
# usage: tools/build_rev.sh <rev> <name>
set -e
rev=$1; name=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" cross-scale-mae_amd/csrc | tar -x -C "$tmp"
objs=""
for f in "$tmp"/cross-scale-mae_amd/csrc/*.hip; do
  o="$tmp/$(basename "$f" .hip).o"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c "$f" -o "$o" &
  objs="$objs $o"
done
wait
mkdir -p "$root/build"
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$root/build/lib_$name.so"
rm -rf "$tmp"
echo "built build/lib_$name.so from $rev"

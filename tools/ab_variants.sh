#!/bin/bash
# Runs ON THE GPU BOX: rebuilds libonepiece_hip.so with each set of extra compiler flags and times the fusion path
# (tools/quick_bench.py N).  Usage: bash tools/ab_variants.sh N "flags of variant 1" "flags of variant 2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; shift
cd $R/onepiece_amd/csrc
for V in "$@"; do
  make -B -j8 EXTRA="$V" > /tmp/ab_make.log 2>&1 || { echo "variant [$V]: build failed"; tail -5 /tmp/ab_make.log; continue; }
  echo "variant [$V]"; python $R/tools/quick_bench.py $N 2>&1 | tail -3 | grep -v algorithmic
done

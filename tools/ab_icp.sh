#!/bin/bash
# Runs ON THE GPU BOX: rebuilds the library with each set of extra flags and times the ICP loop (prof_driver icp mode).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && python $R/tools/dump_frames.py /tmp/fi.bin 2 0 > /dev/null
cd $R/onepiece_amd/csrc
for V in "$@"; do
  make -B -j8 EXTRA="$V" > /tmp/ab_make.log 2>&1 || { echo "variant [$V]: build failed"; tail -5 /tmp/ab_make.log; continue; }
  echo "variant [$V]"; $R/tools/prof_driver.bin /tmp/fi.bin 3 0.005 icp | tail -3
done

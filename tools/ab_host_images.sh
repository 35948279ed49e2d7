#!/bin/bash
# Runs ON THE GPU BOX: rebuilds the library with each set of extra flags and times the host-image fusion path.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && python $R/tools/dump_frames.py /tmp/fh.bin 200 0 > /dev/null
cd $R/onepiece_amd/csrc
for V in "$@"; do
  make -B EXTRA="$V" > /tmp/ab_make.log 2>&1 || { echo "variant [$V]: build failed"; continue; }
  echo "variant [$V]"; $R/tools/prof_driver.bin /tmp/fh.bin 3 0.005 host | tail -2
done

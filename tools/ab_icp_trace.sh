#!/bin/bash
# Runs ON THE GPU BOX: per-iteration host trace (launch call / wait / solve) of the ICP loop for each set of extra flags.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && python $R/tools/dump_frames.py /tmp/fi.bin 2 0 > /dev/null
cd $R/onepiece_amd/csrc
for V in "$@"; do
  make -B EXTRA="-DICP_TRACE $V" > /tmp/ab_make.log 2>&1 || { echo "variant [$V]: build failed"; continue; }
  echo "variant [$V]"; $R/tools/prof_driver.bin /tmp/fi.bin 2 0.005 icp 2>&1 | grep "host trace" | tail -4
done

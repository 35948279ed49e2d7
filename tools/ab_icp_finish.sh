#!/bin/bash
# Runs ON THE GPU BOX: times the reference-order finish of op_icp_run (ICP_TRACE build) for each set of extra flags.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && python $R/tools/dump_frames.py /tmp/fi.bin 2 0 > /dev/null
cd $R/onepiece_amd/csrc
for V in "$@"; do
  make -B EXTRA="-DICP_TRACE $V" > /tmp/ab_make.log 2>&1 || { echo "variant [$V]: build failed"; continue; }
  echo "variant [$V]"; $R/tools/prof_driver.bin /tmp/fi.bin 1 0.005 icp 2>&1 | grep "finish trace" | python3 -c "
import sys, statistics
a=[]; b=[]
for l in sys.stdin:
    w=l.split(); a.append(float(w[9])); b.append(float(w[15]))
print('rows to host + first pass: median %.1f us (min %.1f), second pass + fit: median %.1f us, n=%d' % (statistics.median(a), min(a), statistics.median(b), len(a)))"
done

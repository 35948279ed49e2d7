#!/bin/bash
# rocprofv3 kernel stats of the dense tracker; keeps only the stats CSV under gpurun_out/.
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_track
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_track -o track -- python $REPO/tools/track_bench.py 100 > $REPO/gpurun_out/track_prof.log 2>&1 || true
find /tmp/prof_track -name "*kernel_stats.csv" -exec cp {} $REPO/gpurun_out/track_kernel_stats.csv \;
head -30 $REPO/gpurun_out/track_kernel_stats.csv

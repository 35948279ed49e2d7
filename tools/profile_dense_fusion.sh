#!/bin/bash
# rocprofv3 kernel stats of the tracking+fusion pipeline (config 4); keeps only the stats CSV.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_df
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_df -o df -- python $REPO/tools/dense_fusion_bench.py 100 1 > $REPO/gpurun_out/df_prof.log 2>&1 || true
find /tmp/prof_df -name "*kernel_stats.csv" -exec cp {} $REPO/gpurun_out/dense_fusion_kernel_stats.csv \;

#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the default bench.py command; keeps only
# the small per-kernel summary (the full trace is ~70 MB).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python $R/bench.py --timed-only > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
find /tmp/prof_bench -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
tail -2 $OUT/bench_under_rocprof.err

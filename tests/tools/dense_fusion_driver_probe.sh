#!/bin/bash
# Runs ON THE GPU BOX: where does examples/cpp/DenseFusion.bin spend its time?  Writes a 160-frame sequence, runs the driver preloaded with
# 1 and 4 pairs in flight (its own host-side breakdown is in the JSON), then once under rocprofv3 --hip-trace --stats for the API summary.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/df_probe; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python - <<PY
import sys; sys.path.insert(0, "$R")
from onepiece_amd import sequence as Q, synthetic as S
fr = [S.room_frame(300 + i) for i in range(160)]
Q.WriteImageSequence("/tmp/df_seq", [f[0] for f in fr], [f[1] for f in fr], [f[2] for f in fr], 1000.0)
PY
for K in 1 4 4; do env -u GPU_MAX_HW_QUEUES $R/examples/cpp/DenseFusion.bin /tmp/df_seq --voxel 0.01 --pipeline $K --preload --repeat 3 | tail -1; done | tee $OUT/driver.txt
rm -rf /tmp/df_prof; env -u GPU_MAX_HW_QUEUES timeout 300 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/df_prof -o p -- $R/examples/cpp/DenseFusion.bin /tmp/df_seq --voxel 0.01 --pipeline 4 --preload > $OUT/prof.log 2>&1
find /tmp/df_prof -name '*hip_api_stats.csv' -exec cp {} $OUT/hip_api_stats.csv \;
find /tmp/df_prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
head -12 $OUT/hip_api_stats.csv | cut -c1-150

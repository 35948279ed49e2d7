#!/bin/bash
# Runs ON THE GPU BOX: ICP replicas in flight (tools/prof_driver.bin icpk=K): aggregate iteration rates for K = 1, 2, 4, 8 contexts and the
# rocprofv3 kernel stats of k_icp_iter alone (K = 1) and four at a time (K = 4) -> gpurun_out/$TAG.txt
TAG=${TAG:-r05_icp_replicas}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG.txt
cd /tmp && export TMPDIR=/tmp
python $R/tools/dump_frames.py /tmp/icp_frames.bin 20 0 1 > /dev/null
: > $OUT
for K in 1 2 4 8; do $R/tools/prof_driver.bin /tmp/icp_frames.bin 3 0.005 icpk=$K | tail -1 >> $OUT; done
for K in 1 4; do
  rm -rf /tmp/icpk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icpk -o p -- $R/tools/prof_driver.bin /tmp/icp_frames.bin 3 0.005 icpk=$K > /dev/null 2>&1
  echo "--- rocprofv3 --kernel-trace --stats, $K context(s) in flight" >> $OUT
  python - >> $OUT <<PY
import csv, glob
for f in glob.glob('/tmp/icpk/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_icp_iter' in r['Name']:
            print('k_icp_iter: calls %s, average %.1f us, min %.1f us, max %.1f us' % (r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
cat $OUT

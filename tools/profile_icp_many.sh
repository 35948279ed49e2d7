#!/bin/bash
# Runs ON THE GPU BOX: the fp64-mode ICP replicas under rocprofv3 --kernel-trace --stats for K = 1, 2, 4, 8 contexts (tools/icp_many_probe.py ... fp64): wall-clock rates
# (op_icp_run_many = one submitter, at most four iterations in flight; independent runs = a submitter thread per context) and k_icp_iter's average duration per K --
# the chip-side reason the aggregate stops near 2.5 - 3 x: one launch fills more than half of the wave slots, so two overlapping launches already fill the chip.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/icp_many
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for K in 1 2 4 8; do
  python $R/tools/icp_many_probe.py $K 60 fp64 2>/dev/null | tail -1
  rm -rf /tmp/icpm_$K
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icpm_$K -o p -- python $R/tools/icp_many_probe.py $K 60 fp64 > /dev/null 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/icpm_$K/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_icp_iter<1' in r['Name']:
            print('   K = $K under the tracer: k_icp_iter<1, true> calls %s avg %.1f us min %.1f max %.1f' % (r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done

#!/bin/bash
# rocprofv3 kernel stats of the ICP loop (tools/icp_bench.py); keeps only the stats CSV.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_icp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_icp -o icp -- python $REPO/tools/icp_bench.py > $REPO/gpurun_out/icp_prof.log 2>&1 || true
find /tmp/prof_icp -name "*kernel_stats.csv" -exec cp {} $REPO/gpurun_out/icp_kernel_stats.csv \;
python - <<PY
import csv
for r in list(csv.DictReader(open("$REPO/gpurun_out/icp_kernel_stats.csv")))[:12]:
    print(r["Name"][:70].replace("(anonymous namespace)::",""), r["Calls"], "avg %.1f min %.1f max %.1f us  %s%%"%(float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3,float(r["MaxNs"])/1e3,r["Percentage"]))
PY

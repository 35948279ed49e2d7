#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): collects the rocprofv3 kernel stats + PMC counters for the
# fusion path with the torch-free driver, keeps only small summaries under gpurun_out/.
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/dump_frames.py /tmp/frames.bin ${NFRAMES:-100} 0
$R/tools/prof_driver.bin /tmp/frames.bin 3 > $OUT/driver_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o s -- $R/tools/prof_driver.bin /tmp/frames.bin 3 > $OUT/stats_run.log 2>&1
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$T -o p -- $R/tools/prof_driver.bin /tmp/frames.bin 1 > $OUT/pmc_$T.log 2>&1
  F=$(find /tmp/pmc_$T -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then
    python - "$F" "$OUT/pmc_$T.summary.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (r.get("Kernel_Name", "?")[:60], r.get("Counter_Name", "?"))
    agg[k][0] += 1; agg[k][1] += float(r.get("Counter_Value", 0) or 0)
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,dispatches,sum,mean_per_dispatch\n")
    for (k, c), (n, s) in sorted(agg.items()):
        f.write('"%s",%s,%d,%.6g,%.6g\n' % (k, c, n, s, s / n))
PY
  fi
done
ls -la $OUT

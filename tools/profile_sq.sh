#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the fusion kernels with the torch-free driver.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -f /tmp/frames.bin ] || python $R/tools/dump_frames.py /tmp/frames.bin ${NFRAMES:-64} 0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-60)
  rm -rf /tmp/pmc_$T
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$T -o p -- $R/tools/prof_driver.bin /tmp/frames.bin 1 > $OUT/sq_$T.log 2>&1
  F=$(find /tmp/pmc_$T -name '*counter_collection.csv' | head -1)
  if [ -n "$F" ]; then
    python - "$F" "$OUT/sq_$T.summary.csv" <<'PY'
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
  else tail -3 $OUT/sq_$T.log; fi
done
grep -h "k_integrate" $OUT/sq_*.summary.csv

#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats + PMC counters (separate passes, --kernel-trace only) for the ICP loop
# through the torch-free driver (tools/prof_driver.cpp "icp" mode); keeps only per-kernel summaries.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_icp
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc -O2 -I $R/include $R/tools/prof_driver.cpp -L $R/onepiece_amd -lonepiece_hip -Wl,-rpath,$R/onepiece_amd -o /tmp/prof_driver.bin || exit 1
python $R/tools/dump_frames.py /tmp/frames_i.bin 2 0
/tmp/prof_driver.bin /tmp/frames_i.bin 3 0.005 icp > $OUT/driver_plain.log 2>&1
tail -2 $OUT/driver_plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pic_stats -o s -- /tmp/prof_driver.bin /tmp/frames_i.bin 2 0.005 icp > $OUT/stats_run.log 2>&1
find /tmp/pic_stats -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats.csv \;
for C in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  T=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pic_$T -o p -- /tmp/prof_driver.bin /tmp/frames_i.bin 1 0.005 icp > $OUT/pmc_$T.log 2>&1
  F=$(find /tmp/pic_$T -name '*counter_collection.csv' | head -1)
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
  else
    tail -3 $OUT/pmc_$T.log
  fi
done
ls $OUT

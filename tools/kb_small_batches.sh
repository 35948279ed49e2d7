R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python $R/tools/dump_frames.py /tmp/frames.bin 96 0 > /dev/null
for V in "-DKB_VOTE=0" ""; do
  (cd $R/onepiece_amd/csrc && make -B EXTRA="$V" > /tmp/ab_make.log 2>&1) || { tail -5 /tmp/ab_make.log; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I $R/include $R/tools/prof_driver.cpp -L $R/onepiece_amd -lonepiece_hip -Wl,-rpath,$R/onepiece_amd -o $R/tools/prof_driver.bin
  for B in 4 8 16; do
    rm -rf /tmp/kbs; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kbs -o p -- $R/tools/prof_driver.bin /tmp/frames.bin 2 0.005 batch=$B > /dev/null 2>&1
    echo "== build [$V] batch=$B"
    python - $(find /tmp/kbs -name '*kernel_stats.csv' | head -1) <<PYEOF
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_integrate", "k_select", "k_prepare")):
        print("   %-28s calls %3s  avg %9.1f us" % (n.split("(")[0].split("::")[-1][:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PYEOF
  done
done
(cd $R/onepiece_amd/csrc && make -B > /dev/null 2>&1)

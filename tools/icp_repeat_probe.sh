#!/bin/bash
# Runs ON THE GPU BOX: what an ICP iteration's neighbour search costs with a WARM L2.  k_icp_iter is built with the search repeated ICP_REPEAT times inside
# one launch (make EXTRA=-DICP_REPEAT=n; the later passes depend on the first and find this launch's lines in the XCD's L2); the difference between
# the kernel's durations at n and n - 1 is the price of a warm pass -- the ceiling of any scheme that keeps the loop inside one launch.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
[ -f /tmp/frames2.bin ] || python $R/tools/dump_frames.py /tmp/frames2.bin 2 0 > /dev/null
for N in 1 2 3; do
  (cd $R/onepiece_amd/csrc && make -B EXTRA="-DICP_REPEAT=$N" > /tmp/icp_make.log 2>&1) || { tail -5 /tmp/icp_make.log; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I $R/include $R/tools/prof_driver.cpp -L $R/onepiece_amd -lonepiece_hip -Wl,-rpath,$R/onepiece_amd -o $R/tools/prof_driver.bin
  rm -rf /tmp/icp_rep_$N; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/icp_rep_$N -o p -- $R/tools/prof_driver.bin /tmp/frames2.bin 3 0.005 icp > /tmp/icp_rep_$N.txt 2>&1
  echo "== ICP_REPEAT=$N"; grep -E "it/s|iterations" /tmp/icp_rep_$N.txt | tail -2
  python - $(find /tmp/icp_rep_$N -name '*kernel_stats.csv' | head -1) <<PYEOF
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_icp_iter" in r["Name"]:
        print("   %-40s calls %4s  avg %8.2f us  min %8.2f  max %8.2f" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PYEOF
done
(cd $R/onepiece_amd/csrc && make -B > /dev/null 2>&1)

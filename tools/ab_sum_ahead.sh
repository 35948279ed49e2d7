#!/bin/bash
# Runs ON THE GPU BOX.  Sum-form k_integrate: gathers of one (default) or two (-DKC_SUM_AHEAD=2) frames in flight while a frame is applied.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python $R/tools/dump_frames.py /tmp/frames.bin 320 0 > /dev/null
for V in "" "-DKC_SUM_AHEAD=2" "" "-DKC_SUM_AHEAD=2"; do
  (cd $R/onepiece_amd/csrc && make -B EXTRA="$V" > /tmp/ab_make.log 2>&1) || { tail -5 /tmp/ab_make.log; exit 1; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I $R/include $R/tools/prof_driver.cpp -L $R/onepiece_amd -lonepiece_hip -Wl,-rpath,$R/onepiece_amd -o $R/tools/prof_driver.bin
  rm -rf /tmp/abs; PD_UPDATE=sum_form timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abs -o p -- $R/tools/prof_driver.bin /tmp/frames.bin 2 0.005 batch=32 > /dev/null 2>&1
  echo "== [$V]"
  python - $(find /tmp/abs -name '*kernel_stats.csv' | head -1) <<PYEOF
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_integrate" in r["Name"]:
        print("   k_integrate calls %s avg %.1f us" % (r["Calls"], float(r["AverageNs"]) / 1e3))
PYEOF
  QB_UPDATE=sum_form timeout 300 python $R/tools/quick_bench.py 1000 2>&1 | grep -E "^rep 2" | cut -c1-50
done
(cd $R/onepiece_amd/csrc && make -B EXTRA="-DKC_SUM_AHEAD=2" > /dev/null 2>&1)
cd $R && timeout 600 python -m pytest tests/test_integration_gpu.py -q -m gpu -k "sum_form" 2>&1 | tail -2
(cd $R/onepiece_amd/csrc && make -B > /dev/null 2>&1)

#!/bin/bash
# Per-level kernel durations of the tracker (one pair at a time, torch-free driver): aggregates the rocprofv3
# kernel trace by kernel name and grid size on the box and prints only the summary.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
hipcc -O2 -I $R/include $R/tools/prof_driver.cpp -L $R/onepiece_amd -lonepiece_hip -Wl,-rpath,$R/onepiece_amd -o /tmp/prof_driver.bin || exit 1
python $R/tools/dump_frames.py /tmp/frames_t.bin 30 300 > /dev/null
rm -rf /tmp/ptl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl -o t -- /tmp/prof_driver.bin /tmp/frames_t.bin 2 0.005 track > /dev/null 2>&1
F=$(find /tmp/ptl -name '*kernel_trace.csv' | head -1)
python - "$F" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:28]
    if not name.startswith(("void k_track", "k_track", "k_prep", "k_norm", "k_emit")): continue
    g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
    agg[(name, g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, g), v in sorted(agg.items()):
    v.sort()
    print("%-30s grid %8d  n %5d  median %7.1f us  p90 %7.1f us" % (name, g, len(v), v[len(v)//2] / 1e3, v[int(len(v)*0.9)] / 1e3))
PY

#!/bin/bash
# Runs ON THE GPU BOX: rebuilds raycast.o with each set of extra compiler flags and runs tools/ops_driver.bin raycast raycast_nc on the
# 250 x 4 room volume, wall clock + rocprofv3 kernel stats.  Usage: bash tools/ab_raycast.sh "flags of variant 1" "flags 2" ...  ("" = defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}
[ -f /tmp/ops_frames.bin ] || python $R/tools/dump_frames.py /tmp/ops_frames.bin ${NFRAMES:-250} 0 ${STRIDE:-4} > /dev/null
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  touch $R/onepiece_amd/csrc/raycast.hip
  (cd $R/onepiece_amd/csrc && make -j8 EXTRA="$V" > /tmp/ab_make.log 2>&1) || { echo "variant [$V]: build failed"; tail -5 /tmp/ab_make.log; continue; }
  echo "variant [$V]"
  $R/tools/ops_driver.bin /tmp/ops_frames.bin 0.005 ${REPS:-5} raycast raycast_nc 2>&1 | grep -v fused | cut -c1-400
  rm -rf /tmp/abrc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abrc -o p -- $R/tools/ops_driver.bin /tmp/ops_frames.bin 0.005 3 raycast raycast_nc > /dev/null 2>&1
  python - <<PY
import csv, glob
for f in glob.glob('/tmp/abrc/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_rc_' in r['Name'] or 'k_raycast' in r['Name']:
            print('   %-14s calls %4s avg %8.1f us  min %8.1f  max %8.1f' % (r['Name'].split('::')[1].split('(')[0], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
touch $R/onepiece_amd/csrc/raycast.hip; (cd $R/onepiece_amd/csrc && make -j8 > /dev/null 2>&1)

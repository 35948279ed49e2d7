#!/bin/bash
# Runs ON THE GPU BOX: kernel durations of the ICP loop with and without the tie marking (rocprofv3 --kernel-trace --stats of prof_driver.bin icp).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/tools/dump_frames.py /tmp/fi.bin 2 100 1 > /dev/null
for T in 0 1; do
  rm -rf /tmp/kt$T
  PD_ICP_TIES=$T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$T -o p -- $R/tools/prof_driver.bin /tmp/fi.bin 3 0.005 icp > /dev/null 2>&1
  echo "PD_ICP_TIES=$T"
  find /tmp/kt$T -name '*kernel_stats.csv' -exec grep k_icp_iter {} \; | sed -E 's/\(anonymous namespace\):://; s/\(float const.*\)",/",/' | cut -c1-160
done

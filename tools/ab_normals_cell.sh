R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && python $R/tools/dump_frames.py /tmp/fi.bin 2 0 > /dev/null
cd $R/onepiece_amd/csrc
for D in 400 300 250 200 150; do
  sed -i "s/OP_TRY(icp_create(xyz, nullptr, n, 0.0, [0-9.]*, mem, device, \&c));/OP_TRY(icp_create(xyz, nullptr, n, 0.0, $D.0, mem, device, \&c));/" icp_grid.hip
  make > /tmp/ab_make.log 2>&1 || { echo "build failed"; continue; }
  echo "divisor $D: $($R/tools/prof_driver.bin /tmp/fi.bin 1 0.005 icp | head -3 | tail -1 | cut -d, -f1)"
done

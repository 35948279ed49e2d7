cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/tools/dump_frames.py /tmp/fi.bin 2 100 1 > /dev/null
for rep in 1 2 3; do
for T in 0 1; do echo "PD_ICP_TIES=$T"; PD_ICP_TIES=$T $R/tools/prof_driver.bin /tmp/fi.bin 3 0.005 icp | tail -2; done
done

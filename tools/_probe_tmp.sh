R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
python $R/tools/dump_frames.py /tmp/frames.bin 96 0 > /dev/null
for V in "-DKC_CHUNK=32" "-DKC_CHUNK=64" "-DKC_CHUNK=128" "-DKC_CHUNK=256"; do
  (cd $R/onepiece_amd/csrc && make -B EXTRA="$V" > /tmp/ab_make.log 2>&1) || { echo "variant [$V]: build failed"; continue; }
  echo "variant [$V]"
  for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    T=$(echo $C | tr ' ' '_'); rm -rf /tmp/pc_$T
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pc_$T -o p -- $R/tools/prof_driver.bin /tmp/frames.bin 1 0.005 batch=32 > /dev/null 2>&1
    python $R/tools/pmc_summary.py /tmp/pc_$T /tmp/pc_$T/sum > /dev/null 2>&1; grep k_integrate /tmp/pc_$T/sum.pmc.csv
  done
  for rep in 1 2; do (cd $R && python bench.py --timed-only --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_launch']; print('  frames/s %.0f KC %.1f us' % (d['value'], k['integrate']*1e3))"); done
done
(cd $R/onepiece_amd/csrc && make -B > /tmp/ab_make.log 2>&1)

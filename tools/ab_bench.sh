#!/bin/bash
# Runs ON THE GPU BOX: rebuilds libonepiece_hip.so with each set of extra compiler flags and runs the bench's timed region
# (python bench.py --timed-only --steps 20 --warmup 3) three times per variant.  Usage: bash tools/ab_bench.sh "flags 1" "flags 2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/onepiece_amd/csrc
for V in "$@"; do
  make -B -j8 EXTRA="$V" > /tmp/ab_make.log 2>&1 || { echo "variant [$V]: build failed"; tail -5 /tmp/ab_make.log; continue; }
  echo "variant [$V]"
  for rep in 1 2 3; do
    (cd $R && python bench.py --timed-only --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_launch']; print('  frames/s %.0f (fusion only %.0f) ms/step %.4f  KA %.1f KB %.1f KC %.1f us' % (d['value'], d['fusion_only_frames_per_s'], d['ms_per_step'], k['prepare_frames']*1e3, k['select']*1e3, k['integrate']*1e3), d['pool'])")
  done
done
make -B -j8 > /tmp/ab_make.log 2>&1

#!/bin/bash
# Runs ON THE GPU BOX: k_select_union's per-workgroup phase times (make EXTRA="-DKB_TRACE ...") on 32-frame batches of the bench scene.
# usage: EXTRA="-DKB_UNION_CHUNK=1" bash tools/kb_trace.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
[ -f /tmp/frames.bin ] || python $R/tools/dump_frames.py /tmp/frames.bin 96 0 > /dev/null
(cd $R/onepiece_amd/csrc && make -B -j8 EXTRA="-DKB_TRACE $EXTRA" > /tmp/kb_make.log 2>&1) || { tail -5 /tmp/kb_make.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I $R/include $R/tools/prof_driver.cpp -L $R/onepiece_amd -lonepiece_hip -Wl,-rpath,$R/onepiece_amd -o $R/tools/prof_driver.bin
echo "== EXTRA=[$EXTRA]"; $R/tools/prof_driver.bin /tmp/frames.bin 2 0.005 batch=32 2>&1 | grep -E "kb trace|frames/s" | tail -3
(cd $R/onepiece_amd/csrc && make -B -j8 > /dev/null 2>&1)

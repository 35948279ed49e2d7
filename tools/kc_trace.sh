#!/bin/bash
# Runs ON THE GPU BOX: where a workgroup of k_integrate spends its time (make EXTRA=-DKC_TRACE; per-wave s_memtime stamps, dumped when the volume is destroyed).
# usage: [PD_UPDATE=sum_form] bash tools/kc_trace.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
[ -f /tmp/frames.bin ] || python $R/tools/dump_frames.py /tmp/frames.bin 96 0 > /dev/null
(cd $R/onepiece_amd/csrc && make -B -j8 EXTRA="-DKC_TRACE $EXTRA" > /tmp/kc_make.log 2>&1) || { tail -5 /tmp/kc_make.log; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I $R/include $R/tools/prof_driver.cpp -L $R/onepiece_amd -lonepiece_hip -Wl,-rpath,$R/onepiece_amd -o $R/tools/prof_driver.bin
for U in "" sum_form; do echo "== update=[$U] EXTRA=[$EXTRA]"; PD_UPDATE=$U $R/tools/prof_driver.bin /tmp/frames.bin 2 0.005 batch=32 2>&1 | grep -E "kc trace|shader cycles per launch" | tail -3; done
(cd $R/onepiece_amd/csrc && make -B -j8 > /dev/null 2>&1)

#!/bin/bash
# Runs ON THE GPU BOX.  A/B of the selection step: every frame claiming its blocks directly (k_select, -DKB_VOTE=0) against the default
# (k_select_vote + k_select_merge): per-kernel times under the tracer (32-frame launches of the bench scene) and the 1000-frame rate.
# Other variants: "VARIANTS='-DX=1|-DY=2' bash tools/ab_kb_select.sh" (an empty entry = the default build).
# usage: bash tools/ab_kb_select.sh [frames=96]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-96}
OUT=$R/gpurun_out/ab_kb_select
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/dump_frames.py /tmp/frames.bin $N 0 > /dev/null
build() { (cd $R/onepiece_amd/csrc && make -B -j8 EXTRA="$1" > /tmp/ab_make.log 2>&1) || { echo "build failed: $1"; tail -5 /tmp/ab_make.log; return 1; }
          /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I $R/include $R/tools/prof_driver.cpp -L $R/onepiece_amd -lonepiece_hip -Wl,-rpath,$R/onepiece_amd -o $R/tools/prof_driver.bin; }
one() { # tag
  local tag=$1
  rm -rf /tmp/ab_$tag; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$tag -o p -- $R/tools/prof_driver.bin /tmp/frames.bin 2 0.005 batch=32 > /dev/null 2>&1
  find /tmp/ab_$tag -name '*kernel_stats.csv' -exec cp {} $OUT/$tag.kernel_stats.csv \;
  echo "== $tag"; python - $OUT/$tag.kernel_stats.csv <<PYEOF
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("k_integrate", "k_select", "k_prepare")):
        print("   %-28s calls %3s  avg %9.1f us  min %8.1f  max %8.1f" % (n.split("(")[0].split("::")[-1][:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PYEOF
  true
  for M in "" sum_form; do QB_UPDATE=$M timeout 300 python $R/tools/quick_bench.py 1000 2>&1 | grep -E "^rep 2" | cut -c1-50; done
}
IFS='|' read -ra VS <<< "${VARIANTS:--DKB_VOTE=0|default}"
i=0
for V in "${VS[@]}"; do [ "$V" = default ] && V=""; build "$V" && one "v${i}_$(echo "$V" | tr -c 'A-Za-z0-9=\n' '_')"; i=$((i+1)); done
build "" > /dev/null

#!/bin/bash
# Runs ON THE GPU BOX.  Round-4 A/B of k_integrate / k_select: one batch list (KC_BANDS=0, the default) against eight image-band lists (KC_BANDS=1), exact and
# sum-form update; per variant the torch-free driver's launch time under the tracer and FETCH_SIZE / WRITE_SIZE per launch.
# usage: bash tools/ab_r04.sh [frames=96]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-96}
OUT=$R/gpurun_out/ab_r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/dump_frames.py /tmp/frames.bin $N 0 > /dev/null
build() { (cd $R/onepiece_amd/csrc && make -B -j8 EXTRA="$1" > /tmp/ab_make.log 2>&1) || { echo "build failed: $1"; tail -5 /tmp/ab_make.log; return 1; }
          /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I $R/include $R/tools/prof_driver.cpp -L $R/onepiece_amd -lonepiece_hip -Wl,-rpath,$R/onepiece_amd -o $R/tools/prof_driver.bin; }
one() { # tag, update mode
  local tag=$1 upd=$2
  PD_UPDATE=$upd $R/tools/prof_driver.bin /tmp/frames.bin 3 0.005 batch=32 > $OUT/$tag.driver.txt 2>&1
  rm -rf /tmp/ab_$tag; PD_UPDATE=$upd timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$tag -o p -- $R/tools/prof_driver.bin /tmp/frames.bin 2 0.005 batch=32 > /dev/null 2>&1
  find /tmp/ab_$tag -name '*kernel_stats.csv' -exec cp {} $OUT/$tag.kernel_stats.csv \;
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/ab_${tag}_$C; PD_UPDATE=$upd timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/ab_${tag}_$C -o p -- $R/tools/prof_driver.bin /tmp/frames.bin 1 0.005 batch=32 > /dev/null 2>&1
    python $R/tools/pmc_summary.py /tmp/ab_${tag}_$C $OUT/$tag.$C > /dev/null 2>&1
  done
  echo "== $tag"; tail -1 $OUT/$tag.driver.txt; grep -E "k_integrate|k_select|k_prepare" $OUT/$tag.kernel_stats.csv | cut -d, -f1-4 | sed 's/void (anonymous namespace):://'
  for C in FETCH_SIZE WRITE_SIZE; do grep -h -E "k_integrate" $OUT/$tag.$C.pmc.csv 2>/dev/null | head -2; done
}
build "-DKC_BANDS=1" && { one bands1_exact ""; one bands1_sum sum_form; }
build "" && { one bands0_exact ""; one bands0_sum sum_form; }
for M in "" sum_form; do echo "== quick_bench 1000 frames, update=[$M]"; QB_UPDATE=$M python $R/tools/quick_bench.py 1000 2>&1 | grep -E "^rep" | cut -c1-60; done

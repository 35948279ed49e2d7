#!/bin/bash
# Runs ON THE GPU BOX (gpurun).  Everything the `roofline` object of bench.py cites, as small csv / txt files under
# gpurun_out/$TAG/ (copy into profiles/ with the round prefix):
#   valu_ubench*.txt, issue_costs.json   issue cost per instruction class in shader cycles (s_memtime ticks)
#   calib.*                        tools/hbm_calib.bin: known-traffic kernels under FETCH_SIZE / WRITE_SIZE / TCC_EA0 request counters
#   batch1.*  / batch32.*          tools/prof_driver.bin with ONE frame per k_integrate launch / with full batches:
#                                  --kernel-trace --stats durations, HBM counters, SQ instruction and cycle counters
# Every --pmc group is its own pass (never combined with other trace domains).
TAG=${TAG:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HBM_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum")
SQ_GROUPS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE GRBM_COUNT")
pass() { # name, counters ("" = --stats), command...
  local name=$1 ctr=$2; shift 2
  local T=$(echo ${ctr:-stats} | tr ' ' '_' | cut -c1-48)
  rm -rf /tmp/pp_${name}_$T
  if [ -z "$ctr" ]; then
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_${name}_$T -o p -- "$@" > $OUT/$name.$T.log 2>&1
    find /tmp/pp_${name}_$T -name '*kernel_stats.csv' -exec cp {} $OUT/$name.kernel_stats.csv \;
  else
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pp_${name}_$T -o p -- "$@" > $OUT/$name.$T.log 2>&1
  fi
  python $R/tools/pmc_summary.py /tmp/pp_${name}_$T $OUT/$name.$T
  tail -2 $OUT/$name.$T.log
}
rocprofv3-avail list 2>/dev/null | grep -i -E "TCC_EA0|FETCH|WRITE_SIZE|TCC_REQ|TCC_READ|TCC_WRITE|MALL|HBM" | head -80 > $OUT/avail_tcc.txt
timeout 120 $R/tools/valu_ubench.bin > $OUT/valu_ubench.txt 2>&1
# issue costs in shader cycles: the microbenchmark at k_integrate's occupancy, long enough for stable figures (tools/issue_model.py)
timeout 300 $R/tools/valu_ubench.bin 4096 8 > $OUT/valu_ubench_iters4096.txt 2>&1
python $R/tools/issue_model.py calibrate $OUT/valu_ubench_iters4096.txt $OUT/issue_costs.json > /dev/null
if [ -z "$SKIP_CALIB" ]; then
# ---- known-traffic calibration (1 repetition per kernel under the counters)
timeout 120 $R/tools/hbm_calib.bin 4096 3 > $OUT/calib.timing.txt 2>&1
for C in "${HBM_GROUPS[@]}"; do pass calib "$C" $R/tools/hbm_calib.bin 4096 1; done
fi
# ---- fusion: one frame per launch, and full batches
python $R/tools/dump_frames.py /tmp/frames.bin ${NFRAMES:-96} 0 > /dev/null
$R/tools/prof_driver.bin /tmp/frames.bin 2 0.005 batch=1 > $OUT/batch1.driver.txt 2>&1
$R/tools/prof_driver.bin /tmp/frames.bin 2 0.005 batch=32 > $OUT/batch32.driver.txt 2>&1
pass batch1 "" $R/tools/prof_driver.bin /tmp/frames.bin 2 0.005 batch=1
pass batch32 "" $R/tools/prof_driver.bin /tmp/frames.bin 2 0.005 batch=32
for C in "${HBM_GROUPS[@]}"; do
  pass batch1 "$C" $R/tools/prof_driver.bin /tmp/frames.bin 1 0.005 batch=1
  pass batch32 "$C" $R/tools/prof_driver.bin /tmp/frames.bin 1 0.005 batch=32
done
for C in "${SQ_GROUPS[@]}"; do
  pass batch1 "$C" $R/tools/prof_driver.bin /tmp/frames.bin 1 0.005 batch=1
  pass batch32 "$C" $R/tools/prof_driver.bin /tmp/frames.bin 1 0.005 batch=32
done
cat $OUT/*.log > $OUT/all_logs.txt 2>/dev/null; for B in batch1 batch32; do  # the launch's duration in shader cycles is printed by the driver (op_volume_stats_launches)
  CYC=$(tail -1 $OUT/$B.driver.txt | sed 's/.* \([0-9]*\) shader cycles per launch.*/\1/')
  python $R/tools/issue_model.py model_pmc $OUT/issue_costs.json $OUT $B $CYC > $OUT/$B.issue_model.json
done
rm -f $OUT/*.log
ls $OUT | head -100

#!/bin/bash
# Runs ON THE GPU BOX (gpurun).  rocprofv3 evidence for the volume operations next to the fusion kernels -- the raycaster (row N4),
# GetPointCloud / ExtractTriangleMesh / Transform / TransformNearest (rows I9, N2), EstimateNormals (G2), BilateralFilter (N5) --
# on the volume tools/ops_driver.bin fuses from NFRAMES room frames STRIDE apart (250 x 4 = the whole 1000-frame orbit), 5 mm voxels:
#   $TAG.<group>.driver.txt          the driver's own wall-clock lines and sizes (blocks, rays, hits, points, triangles)
#   $TAG.<group>.kernel_stats.csv    rocprofv3 --kernel-trace --stats
#   $TAG.<group>.<counters>.pmc.csv  one --pmc pass per counter group (never combined with other trace domains), summed per kernel
# Copy what is to be judged from gpurun_out/$TAG/ into profiles/.
TAG=${TAG:-r05_ops}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
NFRAMES=${NFRAMES:-250}
STRIDE=${STRIDE:-4}
REPS=${REPS:-3}
GROUPS_TO_RUN=${GROUPS_TO_RUN:-"raycast ops"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
HBM_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum")
SQ_GROUPS=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT")
pass() { # name, counters ("" = --stats), command...
  local name=$1 ctr=$2; shift 2
  local T=$(echo ${ctr:-stats} | tr ' ' '_' | cut -c1-48)
  rm -rf /tmp/pv_${name}_$T
  if [ -z "$ctr" ]; then
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_${name}_$T -o p -- "$@" > $OUT/$name.$T.log 2>&1
    find /tmp/pv_${name}_$T -name '*kernel_stats.csv' -exec cp {} $OUT/$name.kernel_stats.csv \;
  else
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pv_${name}_$T -o p -- "$@" > $OUT/$name.$T.log 2>&1
  fi
  python $R/tools/pmc_summary.py /tmp/pv_${name}_$T $OUT/$name.$T
  tail -2 $OUT/$name.$T.log
}
python $R/tools/dump_frames.py /tmp/ops_frames.bin $NFRAMES 0 $STRIDE > /dev/null
for G in $GROUPS_TO_RUN; do
  if [ $G = raycast ]; then OPS="raycast raycast_nc"; else OPS="pointcloud mesh transform transform_nn normals bilateral"; fi
  timeout 600 $R/tools/ops_driver.bin /tmp/ops_frames.bin 0.005 $REPS $OPS > $OUT/$G.driver.txt 2>&1
  cat $OUT/$G.driver.txt
  pass $G "" $R/tools/ops_driver.bin /tmp/ops_frames.bin 0.005 $REPS $OPS
  for C in "${HBM_GROUPS[@]}"; do pass $G "$C" $R/tools/ops_driver.bin /tmp/ops_frames.bin 0.005 1 $OPS; done
  if [ $G = raycast ] || [ -n "$SQ_ALL" ]; then for C in "${SQ_GROUPS[@]}"; do pass $G "$C" $R/tools/ops_driver.bin /tmp/ops_frames.bin 0.005 1 $OPS; done; fi
done
rm -f $OUT/*.log
ls $OUT

#!/bin/bash
# Compiles the REFERENCE's own example sources, unedited and where they lie under /root/reference/example, against this
# repository's class surface (host/one_piece) and links them with libone_piece_hip_host.so -- the demonstration that
# "example/ImageSequenceIntegration links against it unchanged" (BASELINE.json north_star).  Outputs go to
# oracle/_ref/examples/ only (git-ignored, travels to the GPU box like every built artefact); no reference source is
# copied anywhere.  The only stand-in on the include path is the headless viewer (tests/cpp/headless/Visualization):
# the OpenGL GUI is out of scope.  Runs in the build container; a no-op where /root/reference is absent.
#   usage: oracle/tools/build_ref_examples.sh [-fsyntax-only]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
REF=${ONEPIECE_REFERENCE:-/root/reference}
if [ ! -d "$REF/example" ]; then echo "no reference tree at $REF: nothing to do"; exit 0; fi
OUT=$ROOT/oracle/_ref/examples
mkdir -p "$OUT"
HOST=$ROOT/host/one_piece
[ -f "$HOST/libone_piece_hip_host.so" ] || make -C "$HOST"
build_one() { # name, sources...
  local name=$1; shift
  if [ "$MODE" = "-fsyntax-only" ]; then
    for src in "$@"; do g++ -std=c++11 -fsyntax-only -I"$HOST" -I"$ROOT/include" -I"$ROOT/tests/cpp/headless" "$src"; done
  else
    g++ -std=c++11 -O2 -I"$HOST" -I"$ROOT/include" -I"$ROOT/tests/cpp/headless" "$@" -o "$OUT/$name.bin" \
        -L"$HOST" -lone_piece_hip_host -L"$ROOT/onepiece_amd" -lonepiece_hip -lz \
        -Wl,-rpath,'$ORIGIN/../../../host/one_piece' -Wl,-rpath,'$ORIGIN/../../../onepiece_amd'
  fi
  echo "built $name"
}
MODE=$1
for ex in ImageIntegration ImageSequenceIntegration ICPTest MergeMultipleSubmaps MCGenerateMesh EstimateNormals ReadRGBD ConvertImageSequenceToPCD ReadPLYPointCloud ReadPLYMesh DenseOdometry SimplifyMeshClustering PruneMesh EigenTest; do
  build_one $ex "$REF/example/$ex.cpp"
done
# example/DenseFusion (named by BASELINE.json's north_star): two translation units, tracking + submap registration + pose-graph optimisation + fusion
build_one DenseFusion "$REF/example/DenseFusion/DenseFusion.cpp" "$REF/example/DenseFusion/DenseSlam.cpp"

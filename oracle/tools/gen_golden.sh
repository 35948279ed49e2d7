#!/bin/sh
# Regenerates tests/golden/eigen_golden.json.  Build-container only: needs /root/reference's vendored
# Eigen 3.3.7 / Sophus headers (third-party dependencies of the reference, used where they lie).
set -e
R=/root/reference/3rdparty
HERE=$(cd "$(dirname "$0")" && pwd)
mkdir -p "$HERE/../_ref"
g++ -std=c++11 -O3 -msse4.2 -fopenmp -w -I$R/Eigen -I$R/Sophus "$HERE/gen_eigen_golden.cpp" -o "$HERE/../_ref/gen_eigen_golden"
"$HERE/../_ref/gen_eigen_golden" "$HERE/../../tests/golden/eigen_golden.json"
echo "wrote tests/golden/eigen_golden.json"
g++ -std=c++11 -O3 -msse4.2 -w -I$R/Eigen -I$R/Sophus "$HERE/gen_odometry_golden.cpp" -o "$HERE/../_ref/gen_odometry_golden"
"$HERE/../_ref/gen_odometry_golden" "$HERE/../../tests/golden/odometry_golden.json"
echo "wrote tests/golden/odometry_golden.json"
g++ -std=c++11 -O3 -msse4.2 -w -I$R/nanoflann/include "$HERE/gen_nanoflann_golden.cpp" -o "$HERE/../_ref/gen_nanoflann_golden"
"$HERE/../_ref/gen_nanoflann_golden" "$HERE/../../tests/golden/nanoflann_golden.json"
echo "wrote tests/golden/nanoflann_golden.json"
g++ -std=c++11 -O2 -w -I/root/reference/src "$HERE/gen_mc_golden.cpp" -o "$HERE/../_ref/gen_mc_golden"
"$HERE/../_ref/gen_mc_golden" "$HERE/../../tests/golden/mc_table_golden.json"
echo "wrote tests/golden/mc_table_golden.json"

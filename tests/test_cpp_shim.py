"""host/onepiece_hip_shim.hpp (the C++ glue a maintainer drops into the reference) compiles against
look-alikes of the reference's types, links the C-ABI library, and -- on a GPU -- reproduces the
reference-run statistics of the survey scene end to end from C++."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "shim_check.bin")


def _build():
    lib = os.path.join(ROOT, "onepiece_amd")
    subprocess.check_call(["g++", "-std=c++11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "host"),
                           os.path.join(ROOT, "tests", "cpp", "shim_check.cpp"), "-L", lib, "-lonepiece_hip",
                           "-Wl,-rpath," + lib, "-o", EXE])


def test_shim_compiles_as_cxx11_and_links(hip):
    _build()  # -std=c++11 like the reference (CMakeLists.txt:149)
    out = subprocess.run([EXE, "--compile-only"], capture_output=True, text=True)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["compiled"] is True


@pytest.mark.gpu
def test_shim_end_to_end_reproduces_reference_run_statistics(hip):
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    anchor = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_wall_anchor.json")))
    assert r["blocks"] == anchor["blocks"] and r["observed"] == anchor["voxels_with_weight"]
    assert int(r["weight_sum"]) == anchor["weight_sum"] and int(r["xor"], 16) == int(anchor["xor_of_key_hashes"], 16)
    assert r["first_list"] == 21146 and r["reupload_blocks"] == anchor["blocks"]
    assert r["icp_rc"] == 0 and r["icp_pairs"] == 4800 and abs(r["icp_tx"] + 0.002) < 2e-4
    # dense tracker through the shim: a frame against itself -> every pixel pairs with itself, pose stays put
    assert r["track_rc"] == 0 and r["track_pairs"] == 640 * 480 and r["track_ok"] == 1
    assert abs(r["track_tx"]) < 1e-5 and r["track_rmse"] < 1e-5 and r["track_first_pair"] == [0, 0, 0, 0]
    # mesh extraction through the shim: the wall is one surface -> hundreds of thousands of triangles, 3 vertices each
    assert r["mesh_rc"] == 0 and r["mesh_triangles"] > 100000 and r["mesh_vertices"] == 3 * r["mesh_triangles"]
    # tool::BilateralFilter through the shim: the smooth wall moves (most at the reflected borders, where the slope of up to
    # 7.5 mm / pixel turns into a kink), but by millimetres
    assert r["filter_rc"] == 0 and 1e-5 < r["filter_max_change"] < 0.02, r


def test_shim_instantiates_with_real_eigen_types(hip):
    """Only where the reference's vendored Eigen exists (build container)."""
    eigen = "/root/reference/3rdparty/Eigen"
    if not os.path.isdir(eigen):
        pytest.skip("vendored Eigen not present on this machine")
    lib = os.path.join(ROOT, "onepiece_amd")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-w", "-msse4.2", "-I", eigen, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "host"),
                           os.path.join(ROOT, "tests", "cpp", "shim_eigen_check.cpp"), "-L", lib, "-lonepiece_hip", "-Wl,-rpath," + lib,
                           "-o", os.path.join(ROOT, "tests", "cpp", "shim_eigen_check.bin")])

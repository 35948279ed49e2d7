"""HBM-traffic and SQ counters of the fusion kernels for one bench step, collected the way MI355X_MICROARCH.md prescribes:
`rocprofv3 --kernel-trace --pmc <one counter group per pass>` on the torch-free driver tools/prof_driver.bin (rocprofv3's
counter collection segfaults under python + torch) fed with a dump of the very frames the bench fuses.  bench.py calls
measure() for its `roofline.traffic` / `roofline.valu` fields; run as a script it prints the JSON.

gfx950 corrections (same guide), CALIBRATED in round 3 on kernels with known traffic in this path's access shapes
(tools/hbm_calib.hip, profiles/r03_calib.*): 4 GiB read with 4, 8 or 16 bytes per lane, as plane rows, or as a read-modify-write
all report FETCH_SIZE = 2 097 1xx KiB = exactly half (TCC_EA0_RDREQ = 4 GiB / 128 B: read requests are 128 B wide and tallied as
64 B) -> doubled for every access width; 4 GiB written report WRITE_SIZE = 4 194 304 KiB (write requests are 64 B) -> exact.
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tools", "prof_driver.bin")

GROUPS = {
    "traffic_read": ["FETCH_SIZE"],
    "traffic_write": ["WRITE_SIZE"],
    "sq_insts": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"],
    "sq_insts2": ["SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"],
    "sq_cycles": ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"],
    "grbm": ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
}


def build_driver():
    if os.path.exists(DRIVER) and os.path.getmtime(DRIVER) >= os.path.getmtime(os.path.join(ROOT, "tools", "prof_driver.cpp")):
        return DRIVER
    lib = os.path.join(ROOT, "onepiece_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "prof_driver.cpp"),
                           "-L", lib, "-lonepiece_hip", "-Wl,-rpath," + lib, "-o", DRIVER])
    return DRIVER


def _short(k):
    """Kernel name of a trace row -> the short name results are filed under (k_select = the per-frame step, k_select / k_select_vote)."""
    for name in ("k_integrate", "k_select_merge", "k_select", "k_prepare_frames"):
        if name in k:
            return name
    return None


def _pass(counters, frames_file, reps, voxel, timeout, batch=None):
    """One rocprofv3 pass -> {kernel short name: {counter: (dispatches, sum)}} + kernel durations if traced."""
    td = tempfile.mkdtemp(prefix="opc_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", td, "-o", "p"]
        if counters:
            cmd += ["--pmc"] + counters
        else:
            cmd += ["--stats"]
        cmd += ["--", DRIVER, frames_file, str(reps), repr(float(voxel))] + (["batch=%d" % batch] if batch else [])
        env = dict(os.environ, TMPDIR="/tmp")
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
        out = {}
        for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "?")
                name = _short(k)
                if name is None:
                    continue
                d = out.setdefault(name, {}).setdefault(r["Counter_Name"], {})
                did = r.get("Dispatch_Id", str(len(d)))
                d[did] = d.get(did, 0.0) + float(r.get("Counter_Value", 0) or 0)   # rows are per dimension instance: sum them per dispatch
        durs = {}
        for f in glob.glob(os.path.join(td, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "?")
                name = _short(k)
                if name:
                    durs.setdefault(name, []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        return out, durs
    finally:
        shutil.rmtree(td, ignore_errors=True)


def measure(frames_file, voxel=0.005, groups=("traffic_read", "traffic_write", "sq_insts", "sq_insts2", "sq_cycles", "grbm"), timeout=180, batch=None):
    """-> {"k_integrate": {...}, "k_select": {...}}: per-launch means of every counter, HBM bytes per launch, durations."""
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    build_driver()
    res = {}
    for g in groups:
        out, durs = _pass(GROUPS[g], frames_file, 1, voxel, timeout, batch)
        for kern, cs in out.items():
            for c, per_dispatch in cs.items():
                vals = list(per_dispatch.values())
                res.setdefault(kern, {})[c] = {"dispatches": len(vals), "mean_per_launch": sum(vals) / max(len(vals), 1)}
        for kern, d in durs.items():
            res.setdefault(kern, {}).setdefault("profiled_launch_us", {})[g] = sum(d) / len(d) / 1e3
    for kern, r in res.items():
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            rd = r["FETCH_SIZE"]["mean_per_launch"] * 1024.0 * 2.0   # KiB, x2 on gfx950 (128 B requests tallied as 64 B)
            wr = r["WRITE_SIZE"]["mean_per_launch"] * 1024.0
            r["hbm_read_bytes_per_launch"], r["hbm_write_bytes_per_launch"], r["hbm_bytes_per_launch"] = rd, wr, rd + wr
    return res


if __name__ == "__main__":
    print(json.dumps(measure(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.005), indent=1))

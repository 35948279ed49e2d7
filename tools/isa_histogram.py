#!/usr/bin/env python3
"""Static opcode histogram of one kernel of onepiece_amd/csrc/*.hip for gfx950 (MI355X).

    python tools/isa_histogram.py integrate.hip k_integrateILb1 [-D...] [--frames 16]

Compiles the file to device assembly (hipcc -S --cuda-device-only, the flags of csrc/Makefile), takes the body of the
first function whose mangled name contains the given substring, and counts instructions per opcode and per issue class
(VALU / transcendental / SALU / VMEM / LDS / branch / waitcnt).  With --frames F the counts are also shown per frame of
the F-times unrolled frame loop.  Cycle weights (wave64 on a SIMD-32, MI355X_MICROARCH.md "Per-instruction cycle
constants" + tools/valu_ubench.hip): plain fp32/int VALU 2 cycles per wave-instruction, transcendentals (v_rcp/v_sqrt/
v_exp/v_log/v_rsq) and v_pk_*_f32 / 64-bit VALU 4.  The weighted sum is a LOWER bound of the VALU issue time of one pass
over the straight-line code (branches skip parts of it at run time; the dynamic count is what SQ_INSTS_VALU reports).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "onepiece_amd", "csrc")


def issue_class(op):
    op = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)   # encoding suffixes are not part of the operation
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "valu_trans", 4
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return "valu_pk_f32", 4
    if op.startswith("v_") and op.endswith(("_f64", "_u64", "_i64", "_b64")):
        return "valu_64", 4
    if op.startswith("v_"):
        return "valu", 2
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "wait", 0
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branch", 0
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem", 0
    if op.startswith("s_"):
        return "salu", 0
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem", 0
    if op.startswith("ds_"):
        return "lds", 0
    return "other", 0


def main():
    args = sys.argv[1:]
    frames = 0
    if "--frames" in args:
        i = args.index("--frames"); frames = int(args[i + 1]); del args[i:i + 2]
    src, needle, extra = args[0], args[1], args[2:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
                               "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)] + extra, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(needle), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ops = collections.Counter()
    for l in lines[start + 1:end]:
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") or l.endswith(":"):
            continue
        ops[l.split()[0]] += 1
    classes, cyc = collections.Counter(), collections.Counter()
    for op, n in ops.items():
        c, w = issue_class(op)
        classes[c] += n
        cyc[c] += n * w
    meta = {k: next((l.split(",")[-1].strip() for l in lines if needle in l and ".set" in l and k in l), "?") for k in (".num_vgpr", ".numbered_sgpr", ".private_seg_size")}
    total = sum(ops.values())
    print("kernel *%s* in %s %s: %d instructions, vgpr %s sgpr %s scratch %s B" % (needle, src, " ".join(extra), total, meta[".num_vgpr"], meta[".numbered_sgpr"], meta[".private_seg_size"]))
    print("class,instructions,issue_cycles_wave64" + (",per_frame" if frames else ""))
    for c, n in classes.most_common():
        print("%s,%d,%d%s" % (c, n, cyc[c], (",%.1f" % (n / frames)) if frames else ""))
    valu = sum(n for c, n in classes.items() if c.startswith("valu"))
    valu_cyc = sum(n for c, n in cyc.items() if c.startswith("valu"))
    print("VALU total,%d,%d%s" % (valu, valu_cyc, (",%.1f instr / %.1f cycles per frame" % (valu / frames, valu_cyc / frames)) if frames else ""))
    print("opcode,count" + (",per_frame" if frames else ""))
    for op, n in ops.most_common(45):
        print("%s,%d%s" % (op, n, (",%.2f" % (n / frames)) if frames else ""))


if __name__ == "__main__":
    main()

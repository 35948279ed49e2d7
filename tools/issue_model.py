#!/usr/bin/env python3
"""Instruction-issue model of k_integrate: what fraction of the SIMDs' issue capacity a launch uses.

k_integrate with full batches is bound by instruction issue, not by HBM, so its roofline is
    frac = sum over instruction classes (wave-instructions executed x issue cost of the class) / (1024 SIMDs x kernel cycles)
with everything in SHADER clock cycles (s_memtime counts them on gfx950: its rate follows the clock -- 2.1 GHz while
k_integrate runs, 1.2 GHz while the pure-VALU microbenchmark has the chip power-throttled -- whereas GRBM_GUI_ACTIVE keeps
counting at ~2.1 GHz; measured in profiles/r03_valu_ubench_*.txt):
  * wave-instructions per class from the SQ counters of the launch (SQ_INSTS_VALU / SALU / SMEM / BRANCH / LDS / VMEM_RD / VMEM_WR,
    rocprofv3 --pmc, tools/counters.py / tools/profile_roofline.sh);
  * the VALU count split over cost classes in the proportions of the kernel's STATIC opcode histogram (tools/isa_histogram.py:
    the hot loop is straight-line code executed once per frame bit, so static and dynamic proportions agree closely);
  * issue costs per class = s_memtime ticks per wave64 instruction and SIMD measured by tools/valu_ubench.hip at 8 waves per SIMD
    (k_integrate's occupancy); they do not depend on how long the microbenchmark runs (64 .. 4096 iterations);
  * kernel cycles = the longest s_memtime span of one of the launch's resident workgroups (op_volume_stats_launches).

    python tools/issue_model.py calibrate <valu_ubench output .txt> <out.json>             # -> issue costs (commit under profiles/)
    python tools/issue_model.py model <costs.json> <counters.json>                          # counters: {"SQ_INSTS_VALU": .., "kernel_cycles": ..}
    python tools/issue_model.py model_pmc <costs.json> <dir> <prefix> <kernel cycles>       # counters from <dir>/<prefix>.*.pmc.csv (tools/pmc_summary.py)
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS = 1024  # 256 CUs x 4

# OP number of tools/valu_ubench.hip -> name of the cost row
UBENCH_OPS = {0: "v_mul_f32", 1: "v_fma_f32", 2: "v_pk_mul_f32", 10: "v_pk_fma_f32", 3: "v_rcp_f32", 4: "v_div_scale_f32", 5: "v_div_fmas_f32",
              6: "v_div_fixup_f32", 7: "v_cndmask_vop2_back_to_back", 8: "v_cmp_f32", 9: "v_mov_b32", 11: "v_cndmask_e64", 17: "v_cndmask_vop2_2src",
              20: "vop2_mix", 12: "v_bfi_b32", 13: "v_min_f32", 14: "v_max_u32", 15: "v_and_b32", 16: "v_add_f32", 18: "v_add_u32", 19: "v_cmp_u64",
              21: "s_add_u32", 22: "s_and_b64", 23: "v_mul_s_add_mixed", 24: "s_nop", 25: "v_fmac_f32", 26: "v_trunc_f32", 27: "v_cvt_i32_f32",
              28: "v_max_i32", 29: "v_lshlrev_b32", 30: "s_mul_i32", 31: "s_waitcnt_idle", 32: "s_cbranch_not_taken", 33: "v_readlane_b32",
              34: "ds_read_b32", 35: "s_load_dword", 41: "v_add_f32_dependent_chain"}


def calibrate(ubench_txt, waves=8):
    """output of tools/valu_ubench.bin -> {row name: shader cycles per wave64 instruction and SIMD} at `waves` waves per SIMD"""
    costs = {}
    for line in open(ubench_txt):
        m = re.search(r"OP\s+(\d+) waves/SIMD (\d+) iters \d+: [\d.]+ s_memtime ticks per wave-instruction -> ([\d.]+) per SIMD-issue slot", line)
        if m and int(m.group(2)) == waves:
            costs[UBENCH_OPS[int(m.group(1))]] = float(m.group(3))
    return costs


def valu_cost_key(op):
    """static opcode (llvm mnemonic) -> row of the cost table"""
    op = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", op)
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "v_rcp_f32"
    if op.startswith(("v_fma_", "v_mad_")):
        return "v_fma_f32"
    if op.startswith("v_fmac"):
        return "v_fmac_f32"
    if op.startswith("v_div_scale"):
        return "v_div_scale_f32"
    if op.startswith("v_div_fmas"):
        return "v_div_fmas_f32"
    if op.startswith("v_div_fixup"):
        return "v_div_fixup_f32"
    if op.startswith("v_cmp"):
        return "v_cmp_u64" if op.endswith(("64",)) else "v_cmp_f32"
    if op.startswith("v_cndmask"):
        return "v_cndmask_e64"       # the kernel's selects are VOP3-encoded or separated by other instructions
    if op.startswith(("v_bfi", "v_bfe", "v_perm", "v_alignbit", "v_lshl_or", "v_and_or", "v_or3", "v_add3", "v_lshl_add", "v_mbcnt")):
        return "v_bfi_b32"
    if op.startswith(("v_min", "v_max", "v_med3")):
        return "v_max_i32" if op.endswith(("i32", "u32")) else "v_min_f32"
    if op.startswith(("v_trunc", "v_floor", "v_ceil", "v_rndne", "v_fract")):
        return "v_trunc_f32"
    if op.startswith("v_cvt"):
        return "v_cvt_i32_f32"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "v_readlane_b32"
    if op.startswith(("v_pk_",)):
        return "v_pk_fma_f32"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64")):
        return "v_fma_f32"
    if op.startswith(("v_mul_f32", "v_mul_u32_u24", "v_mul_i32_i24")):
        return "v_mul_f32"
    if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32")):
        return "v_add_f32"
    if op.startswith(("v_lshlrev", "v_lshrrev", "v_ashrrev")):
        return "v_lshlrev_b32"
    return "v_and_b32"               # mov / and / or / xor / integer add-sub: plain VOP1 / VOP2


def static_valu_mix(kernel_substr="k_integrateILb1ELb1", src="integrate.hip"):
    """-> {cost row: share of the kernel's static VALU instructions} from tools/isa_histogram.py's per-opcode table"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_histogram.py"), src, kernel_substr], capture_output=True, text=True, check=True).stdout
    counts, seen = {}, False
    for line in out.splitlines():
        if line.startswith("opcode,count"):
            seen = True
            continue
        if seen and "," in line:
            op, n = line.split(",")[:2]
            if op.startswith("v_"):
                counts[valu_cost_key(op)] = counts.get(valu_cost_key(op), 0) + int(n)
    tot = float(sum(counts.values()))
    return {k: v / tot for k, v in sorted(counts.items())}


def model(costs, mix, counters):
    """counters: per-launch means {SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_INSTS_SMEM, SQ_INSTS_BRANCH, SQ_INSTS_LDS, SQ_INSTS_VMEM_RD, SQ_INSTS_VMEM_WR} and
    kernel_cycles = the launch's duration in shader cycles"""
    valu_cost = sum(share * costs[row] for row, share in mix.items())
    kernel_cycles = float(counters["kernel_cycles"])
    # SQ_INSTS_SALU counts the scalar ALU instructions; SMEM and branches have their own counters.  The back-to-back costs of s_load and
    # ds_read measured by the microbenchmark contain the wait for their data, which other waves' instructions fill in a real kernel:
    # SMEM, LDS and VMEM instructions are priced at what they take from the issue port, a scalar / plain vector slot.
    classes = {
        "valu": (counters["SQ_INSTS_VALU"], valu_cost),
        "salu": (counters.get("SQ_INSTS_SALU", 0.0), costs["s_add_u32"]),
        "smem": (counters.get("SQ_INSTS_SMEM", 0.0), costs["s_add_u32"]),
        "branch": (counters.get("SQ_INSTS_BRANCH", 0.0), costs.get("s_cbranch_not_taken", costs["s_add_u32"])),
        "lds": (counters.get("SQ_INSTS_LDS", 0.0), costs["v_mov_b32"]),
        "vmem": (counters.get("SQ_INSTS_VMEM_RD", 0.0) + counters.get("SQ_INSTS_VMEM_WR", 0.0), costs["v_mov_b32"]),
    }
    issue = {k: n * c for k, (n, c) in classes.items()}
    total, cap = sum(issue.values()), SIMDS * kernel_cycles
    return {"issue_cycles_per_launch": total, "simd_cycles_per_launch": cap, "frac": total / cap, "kernel_cycles": kernel_cycles,
            "classes": {k: {"wave_instructions": classes[k][0], "cycles_each": classes[k][1], "issue_cycles": issue[k], "share_of_capacity": issue[k] / cap} for k in classes},
            "valu_mix": mix, "valu_cycles_each": valu_cost}


def counters_from_summaries(directory, prefix, kernel="k_integrate"):
    """per-launch means of every counter of `kernel` in <directory>/<prefix>.*.pmc.csv (tools/pmc_summary.py files)"""
    import glob
    out = {}
    for f in glob.glob(os.path.join(directory, prefix + ".*.pmc.csv")):
        for r in csv.DictReader(open(f)):
            if r["kernel"].startswith(kernel):
                out[r["counter"]] = float(r["mean_per_dispatch"])
    return out


def mixed_check(ubench_txt):
    """the MIXED line of tools/valu_ubench.bin: one loop with k_integrate's instruction-class ratio at 8 waves per SIMD, measured against the additive
    model's and the VALU-only prediction from the single-class rows of the same run -> {additive_over_measured, valu_only_over_measured, ...} or None"""
    for line in open(ubench_txt):
        m = re.search(r"MIXED kernel .*measured ([\d.]+) shader cycles.*additive model ([\d.]+) \(VALU ([\d.]+) \+ SALU/SMEM ([\d.]+) \+ branch ([\d.]+)\) = ([\d.]+) x measured; VALU-only ([\d.]+) = ([\d.]+) x measured", line)
        if m:
            g = [float(x) for x in m.groups()]
            return {"measured_cycles_per_iteration": g[0], "additive_model_cycles": g[1], "valu_cycles": g[2], "salu_smem_cycles": g[3], "branch_cycles": g[4],
                    "additive_over_measured": g[5], "valu_only_over_measured": g[7],
                    "additive_model_holds": abs(g[5] - 1.0) <= 0.10,
                    "note": "96 VALU : 24 SALU : 6 branch : 3 SMEM per iteration at 8 waves per SIMD (tools/valu_ubench.hip OP 40); the additive model is used for "
                            "roofline.frac only when it predicts this kernel's time within 10 %; otherwise frac is the VALU-only share"}
    return None


if __name__ == "__main__":
    if sys.argv[1] == "calibrate":
        out = {"unit": "shader cycles (s_memtime ticks) per wave64 instruction and SIMD at 8 waves per SIMD, from " + os.path.basename(sys.argv[2]),
               "costs": calibrate(sys.argv[2]), "valu_mix_k_integrate_plain": static_valu_mix(), "mixed_check": mixed_check(sys.argv[2])}
        json.dump(out, open(sys.argv[3], "w"), indent=1)
        print(json.dumps(out, indent=1))
    elif sys.argv[1] == "model_pmc":
        cj = json.load(open(sys.argv[2]))
        cnt = counters_from_summaries(sys.argv[3], sys.argv[4])
        cnt["kernel_cycles"] = float(sys.argv[5])
        print(json.dumps(model(cj["costs"], cj["valu_mix_k_integrate_plain"], cnt), indent=1))
    else:
        cj = json.load(open(sys.argv[2]))
        print(json.dumps(model(cj["costs"], cj["valu_mix_k_integrate_plain"], json.load(open(sys.argv[3]))), indent=1))

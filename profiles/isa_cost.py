#!/usr/bin/env python3
"""Cycle-weighted instruction budget of one kernel of a hipcc -save-temps assembly file, per basic block.

    make -C neural-gauge-fields_amd/csrc asm
    python profiles/isa_cost.py neural-gauge-fields_amd/csrc/build/asm/ngf_field-hip-amdgcn-amd-amdhsa-gfx950.s \
           'render_kernelINS_14TriPlanePolicyILb1ELb1ELi12ELi1ELb0EEELb1ELb0E' [profiles/r05_micro_valu_cost.txt] [min_cycles]

An instruction COUNT is not a cycle count on gfx950: profiles/micro/valu_cost.hip (output: profiles/r05_micro_valu_cost.txt) measures what a wave64
instruction of each class costs a SIMD that runs three waves (the render kernels' occupancy): ~2.9 cycles for plain fp32 / integer add / logic
ops, ~4.4 for v_max / v_med3 / shifts-with-add / conversions / DPP, ~5.5 for compares and lane reads, ~8.3 for exp / log / rcp, 32 for an fp32
MFMA 16x16x4 (matrix pipe; it shares the SIMD with the vector instructions: the two add up, profiles/r02_micro_mfma_valu_overlap.txt).
This script prices every instruction of a kernel with the measured cost of its class and sums per basic block: the bucket table of DESIGN.md 4.11.
LDS and memory instructions are priced at their ISSUE slot only (their data paths are other units)."""
import re
import sys

# class -> (measured row of the micro-benchmark, fallback cycles)
CLASS_ROWS = {
    "simple": ("v_add_f32", 2.9), "pk": ("v_pk_fma_f32 (2 FMAs / lane)", 4.8), "quarter": ("v_lshl_add_u32", 4.4), "cmp": ("v_cmp_gt_f32 -> vcc", 5.5),
    "trans": ("v_exp_f32", 8.3), "lane": ("v_readlane_b32", 5.5), "swap": ("v_permlane32_swap", 9.4), "cndmask": ("v_cndmask_b32_e64 (s[20:21])", 4.6),
    "dpp": ("v_mul_f32_dpp row_shr", 4.4), "mov": ("v_mov_b32", 2.9),
}
SIMPLE = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
          "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_not_b32", "v_mul_u32_u24", "v_mul_i32_i24", "v_mac_f32")
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32")


def load_costs(path):
    rows = {}
    if path:
        for l in open(path):
            m = re.match(r"^(.{32})\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l.rstrip("\n"))
            if m:
                rows[m.group(1).strip()] = float(m.group(4))          # the three-waves-per-SIMD column
    return {k: rows.get(row, fb) for k, (row, fb) in CLASS_ROWS.items()}


def classify(op):
    if op.startswith("v_mfma_f32_16x16x4"):
        return "mfma", 32.0
    if op.startswith("v_mfma_f32_32x32x2"):
        return "mfma", 64.0
    if op.startswith("v_mfma"):
        return "mfma", 16.0 if "16x16" in op else 32.0
    if op.startswith("v_pk_"):
        return "pk", None
    if "_dpp" in op:
        return "dpp", None
    if op.startswith(("v_cmp", "v_cmpx")):
        return "cmp", None
    if op.startswith("v_cndmask"):
        return "cndmask", None
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane", None
    if op.startswith("v_permlane"):
        return "swap", None
    if op.startswith(TRANS):
        return "trans", None
    if op.startswith(("v_mov_b32", "v_accvgpr")):
        return "mov", None
    if op.startswith(SIMPLE):
        return "simple", None
    if op.startswith("v_"):
        return "quarter", None
    if op.startswith("ds_"):
        return "lds", 0.0
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 0.0
    if op.startswith("s_"):
        return "salu", 0.0          # scalar pipe / wait states of ONE wave: no vector-pipe cycles (the SIMD's other waves issue meanwhile); counted, not priced
    return "other", 0.0


def cost(path, needle, costs_path=None, min_cycles=40.0):
    costs = load_costs(costs_path)
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(needle) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], None
    for i in range(start, end + 1):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m or cur is None:
            cur = {"label": m.group(1) if m else "entry", "cycles": {}, "count": {}}
            blocks.append(cur)
            if m:
                continue
        s = l.strip()
        if not s or s[0] in ";." or s.endswith(":"):
            continue
        op = s.split()[0]
        cls, fixed = classify(op)
        cyc = fixed if fixed is not None else costs.get(cls, 4.4)
        cur["cycles"][cls] = cur["cycles"].get(cls, 0.0) + cyc
        cur["count"][cls] = cur["count"].get(cls, 0) + 1
    print(f"# class costs (cycles per wave64 instruction at three waves per SIMD): " + ", ".join(f"{k} {v:.1f}" for k, v in sorted(costs.items())))
    out = []
    for b in blocks:
        vec = sum(v for k, v in b["cycles"].items() if k not in ("mfma",))
        mf = b["cycles"].get("mfma", 0.0)
        if vec + mf >= min_cycles:
            cnt = " ".join(f"{k}:{b['count'][k]}" for k in sorted(b["count"]))
            print(f"{b['label']:>12}  vector {vec:7.0f} cyc  matrix {mf:6.0f} cyc   [{cnt}]")
        out.append((b["label"], vec, mf, b["count"]))
    return out


if __name__ == "__main__":
    cost(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None, float(sys.argv[4]) if len(sys.argv) > 4 else 40.0)

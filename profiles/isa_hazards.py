#!/usr/bin/env python3
"""Static lint of hipcc -save-temps assembly: packed fp32 VALU results that reach a matrix instruction.

Round 3 found the InfoInv NGF_F_SPLIT_BF16 colour pass timing-dependent (1 launch in 4 200 gave ~1e-3 different layer-1 accumulators) when
hipcc compiled the positional-factor chain to v_pk_mul_f32 / v_pk_add_f32 (op_sel / neg modifiers) + v_pk_mov_b32, and bit-stable with the same
IEEE operations as single VALU instructions.  The mechanism below the ISA is not known, so the pattern is kept OUT of the
shipped kernels mechanically: this script walks every kernel of an assembly file and reports

  direct   a v_pk_{mul,add,fma}_f32 / v_pk_mov_b32 result register read by a v_mfma_* within WINDOW instructions, and
  one hop  such a result read by a VALU instruction whose own result a v_mfma_* reads, both within WINDOW instructions

(`--kernels REGEX` restricts the kernels, `--window N` the distance).  That report is INFORMATIVE: the shipped kernels, stable over 1.2 M launches,
also hold packed results that reach matrix instructions (the 3-term bf16 splits compile to v_pk_add_f32 ... neg), so the pattern alone is not the
defect.  What the ISA diff of the unstable (-DNGF_EXP_PACKED_PE) and the shipped build of the InfoInv split kernel shows (`--forms`,
profiles/r04_isa_packed_diff.txt) is three instruction FORMS that only the unstable build contains:

    v_pk_mov_b32 (any)      v_pk_mul_f32 with crossed halves (op_sel:[0,1] op_sel_hi:[1,0])
    v_pk_add_f32 with crossed halves AND negated second source (op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1])

-- hipcc's code for  sn' = 2 sn cs,  cs' = (cs - sn)(cs + sn)  on (sn, cs) register pairs.  `--lint` (exit status 1 on a hit) forbids these
forms in every kernel that issues bf16 matrix instructions; tests/test_isa_lint.py runs it on the shipped translation units and checks that it
does flag the -DNGF_EXP_PACKED_PE build.  An empirical fence around a defect whose mechanism below the ISA is unknown, not an explanation.
    make -C neural-gauge-fields_amd/csrc asm && python profiles/isa_hazards.py neural-gauge-fields_amd/csrc/build/asm/ngf_field-hip-amdgcn-amd-amdhsa-gfx950.s"""
import argparse
import re
import sys

PK = re.compile(r"^v_pk_(mul|add|fma)_f32|^v_pk_mov_b32")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def parse(line):
    s = line.strip()
    if not s or s[0] in ";." or s.endswith(":") or s.startswith(";;"):
        return None
    s = s.split(";")[0].strip()
    parts = s.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def kernels(path):
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m and i + 1 < len(lines):
            j = i + 1
            while j < len(lines) and "s_endpgm" not in lines[j] and not re.match(r"^_Z\w+:", lines[j]):
                j += 1
            if j < len(lines) and "s_endpgm" in lines[j]:
                yield m.group(1), lines[i + 1:j + 1]
            i = j
        i += 1


def scan(body, window):
    ins = []
    for l in body:
        p = parse(l)
        if p:
            ins.append(p)
    found = []
    npk = 0
    for i, (op, ops) in enumerate(ins):
        if not PK.match(op) or not ops:
            continue
        npk += 1
        w = regs(ops[0])
        live = set(w)
        hop = {}                     # register -> index of the VALU instruction that derived it from the packed result
        for j in range(i + 1, min(i + 1 + window, len(ins))):
            op2, ops2 = ins[j]
            if not ops2:
                continue
            is_mfma = op2.startswith("v_mfma") or op2.startswith("v_smfmac")
            srcs = set().union(*[regs(o) for o in ops2[1:]]) if len(ops2) > 1 else set()
            dst = regs(ops2[0]) if op2.startswith(("v_", "ds_read", "global_load", "buffer_load", "scratch_load")) else set()
            if is_mfma:
                if srcs & live:
                    found.append(("direct", i, j, op, op2, sorted(srcs & live)[:2]))
                hit = [r for r in srcs if r in hop]
                if hit:
                    found.append(("one hop", i, j, op, ins[hop[hit[0]]][0] + " -> " + op2, hit[:2]))
            elif op2.startswith("v_") and srcs & live:
                for r in dst:
                    hop[r] = j
            live -= dst                                       # overwritten
            for r in list(hop):
                if r in dst and hop[r] != j:
                    del hop[r]
            if not live and not hop:
                break
    return npk, found


MODS = re.compile(r"(op_sel\S*|neg_\S+)")


def pk_forms(body):
    """Counter of (opcode, modifiers) of the packed fp32 instructions of a kernel."""
    import collections
    c = collections.Counter()
    for l in body:
        p = parse(l)
        if p and PK.match(p[0]):
            c[(p[0], " ".join(MODS.findall(l)))] += 1
    return c


def forbidden(form):
    op, mods = form
    crossed = "op_sel:[0,1] op_sel_hi:[1,0]" in mods
    return op == "v_pk_mov_b32" or (op == "v_pk_mul_f32" and crossed) or (op == "v_pk_add_f32" and crossed and "neg_lo:[0,1]" in mods and "neg_hi:[0,1]" in mods)


def lint(path, pattern=".*", window=400):
    """[(kernel, form, count)] of the forbidden packed forms whose RESULT FLOWS INTO a bf16 matrix instruction: registers written by such an
    instruction are tainted, a VALU instruction with a tainted source taints its destination, an untainted write clears a register; a hit is
    a v_mfma_*bf16 with a tainted source within `window` instructions (straight-line order).  (Without the data-flow condition the fence would
    also stop at library arithmetic that never meets the matrix pipe -- ocml's powf of the UV tone map compiles to the same forms.)"""
    import collections
    hits = []
    for name, body in kernels(path):
        if not re.search(pattern, name):
            continue
        ins = [q for q in (parse(l) + (l,) if parse(l) else None for l in body) if q]
        if not any(op.startswith("v_mfma") and "bf16" in op for op, _, _ in ins):
            continue
        found = collections.Counter()
        for i, (op, ops, raw) in enumerate(ins):
            form = (op, " ".join(MODS.findall(raw)))
            if not PK.match(op) or not forbidden(form) or not ops:
                continue
            taint = set(regs(ops[0]))
            for j in range(i + 1, min(i + 1 + window, len(ins))):
                op2, ops2, _ = ins[j]
                if not ops2:
                    continue
                srcs = set().union(*[regs(o) for o in ops2[1:]]) if len(ops2) > 1 else set()
                if op2.startswith("v_mfma"):
                    if "bf16" in op2 and srcs & taint:
                        found[form] += 1
                        break
                    taint -= regs(ops2[0]) - (srcs & taint and regs(ops2[0]) or set())
                    continue
                writes = op2.startswith(("v_", "ds_read", "global_load", "buffer_load", "scratch_load")) and not op2.startswith(("v_cmp", "v_cmpx"))
                if writes:
                    dst = regs(ops2[0])
                    if op2.startswith("v_") and srcs & taint:
                        taint |= dst
                    else:
                        taint -= dst
                if not taint:
                    break
        for form, n in sorted(found.items()):
            hits.append((name, form, n))
    return hits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernels", default=".*")
    ap.add_argument("--window", type=int, default=64)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--forms", action="store_true", help="print the packed fp32 instruction forms of the selected kernels")
    ap.add_argument("--lint", action="store_true", help="exit 1 if a kernel with bf16 matrix instructions holds a forbidden packed form")
    a = ap.parse_args()
    if a.forms:
        for name, body in kernels(a.asm):
            if re.search(a.kernels, name):
                print(name[:150])
                for (op, mods), n in sorted(pk_forms(body).items()):
                    print(f"    {n:5d}  {op} {mods}" + ("      <-- forbidden in bf16-MFMA kernels" if forbidden((op, mods)) else ""))
        return 0
    if a.lint:
        hits = lint(a.asm, a.kernels)
        for name, (op, mods), n in hits:
            print(f"{name[:150]}: {n} x {op} {mods}")
        print(f"{len(hits)} forbidden packed form(s) in bf16-MFMA kernels of {a.asm}")
        return 1 if hits else 0
    bad = 0
    for name, body in kernels(a.asm):
        if not re.search(a.kernels, name):
            continue
        npk, found = scan(body, a.window)
        mf = sum(1 for l in body if l.strip().startswith(("v_mfma", "v_smfmac")))
        if found or not a.quiet:
            print(f"{name[:150]}: {npk} packed fp32 instructions, {mf} matrix instructions, {len(found)} packed results reaching a matrix instruction within {a.window}")
        for kind, i, j, op, op2, r in found[:8]:
            print(f"    {kind}: instruction {i} {op} -> {j} ({j - i} later) {op2}  registers {r}")
        bad += len(found)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

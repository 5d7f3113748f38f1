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


# ---- hand-placed hazards of the inline assembly (round 5) ---------------------------------------------------------
# hipcc's hazard recogniser does not look inside an `asm` statement.  Two kinds of hand-written instructions in this library read a register
# that a preceding instruction may still be writing:
#   (1) LDS stores whose DATA operands are accumulator registers (csrc/ngf_uv.hpp: `ds_write2st64_b32 v, a, a` stores a layer's rows straight
#       from the MFMA result registers).  Required distance from the MFMA that writes the register, in wait states (LLVM GCNHazardRecognizer,
#       gfx950: XDL write VGPR -> VALU / memory read = passes + 4 for 4 / 8 / 16-pass instructions; the non-XDL fp32 MFMAs need passes + 2 --
#       the lint asks for the larger figure for every matrix instruction): 8 passes (16x16x4 f32) -> 12, 16 passes (32x32x2 f32) -> 20.
#   (2) DPP instructions (csrc/ngf_render.hpp split_chain: `v_mul_f32_dpp` / `v_add_f32_dpp` in place): a VALU write of a VGPR followed by a DPP
#       read of it needs 2 wait states.
# Wait states are counted as the recogniser does: every instruction between producer and consumer is one, `s_nop N` is N + 1.  The scan walks
# backwards in text order and stops at the kernel's start; a label does not stop it (the stores sit in straight-line code behind their k loop:
# a writer in a predecessor that is not the textual one would be missed -- the GPU parity and determinism tests remain the net for that).
def _wait_states(op, ops):
    if op == "s_nop":
        try:
            return int(ops[0], 0) + 1
        except (ValueError, IndexError):
            return 1
    return 1


def _mfma_required(op):
    m = re.match(r"v_s?mfma\w*?_(\d+)x(\d+)x(\d+)", op)
    if not m:
        return 20
    rows = int(m.group(1))
    return 20 if rows >= 32 else (12 if rows >= 16 else 8)      # 16- / 8- / 4-pass shapes, XDL figures of gfx950 (passes + 4)


AGPR = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")


def _agprs(tok):
    out = set()
    for m in AGPR.finditer(tok):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint_mfma_to_lds(path, pattern=".*", lookback=600):
    """[(kernel, instruction index, store, writer, wait states found, required)] for every LDS store with an AGPR data operand that follows the
    matrix instruction writing that register by fewer wait states than required.  Also returns the number of such stores checked."""
    bad, checked = [], 0
    for name, body in kernels(path):
        if not re.search(pattern, name):
            continue
        ins = [q for q in (parse(l) for l in body) if q]
        for i, (op, ops) in enumerate(ins):
            if not op.startswith("ds_write") or len(ops) < 2:
                continue
            data = set().union(*[_agprs(o) for o in ops[1:]])
            if not data:
                continue
            checked += 1
            need = set(data)
            states = 0
            for j in range(i - 1, max(i - 1 - lookback, -1), -1):
                op2, ops2 = ins[j]
                if op2.startswith(("v_mfma", "v_smfmac")) and ops2 and (_agprs(ops2[0]) & need):
                    req = _mfma_required(op2)
                    if states < req:
                        bad.append((name, i, op + " " + ", ".join(ops), op2 + " " + ops2[0], states, req))
                    need -= _agprs(ops2[0])
                    if not need:
                        break
                elif op2.startswith("v_accvgpr_write") and ops2 and (_agprs(ops2[0]) & need):
                    need -= _agprs(ops2[0])              # a copy into the AGPR: a VALU write, covered by the assembler-visible 1-state rule of ds data
                    if not need:
                        break
                states += _wait_states(op2, ops2)
                if states >= 20:
                    break
    return bad, checked


def lint_valu_to_dpp(path, pattern=".*", required=2):
    """[(kernel, instruction index, dpp instruction, writer, wait states)] for every *_dpp instruction that reads a VGPR written by a VALU
    instruction fewer than `required` wait states earlier.  Also returns the number of DPP instructions checked."""
    bad, checked = [], 0
    for name, body in kernels(path):
        if not re.search(pattern, name):
            continue
        ins = [q for q in (parse(l) for l in body) if q]
        for i, (op, ops) in enumerate(ins):
            if "_dpp" not in op or len(ops) < 2:
                continue
            checked += 1
            # sources: every register operand after the destination -- and the destination itself (bound_ctrl:0 / row masks keep the old value:
            # the in-place forms `v_mul_f32_dpp v1, v1, v2` read it anyway)
            srcs = set().union(*[{r for r in regs(o.split()[0]) if r[0] == "v"} for o in ops[1:] if o and o[0] in "v"]) | {r for r in regs(ops[0]) if r[0] == "v"}
            states = 0
            for j in range(i - 1, max(i - 1 - 8, -1), -1):
                op2, ops2 = ins[j]
                if states >= required:
                    break
                if op2.startswith("v_") and not op2.startswith(("v_cmp", "v_cmpx", "v_nop")) and ops2 and ({r for r in regs(ops2[0]) if r[0] == "v"} & srcs):
                    bad.append((name, i, op + " " + ", ".join(ops), op2 + " " + ops2[0], states))
                    break
                states += _wait_states(op2, ops2)
    return bad, checked


def lint_accvgpr_write_in_exec_regions(path, pattern=".*"):
    """the one build of uv_render_kernel that produced wrong densities (17 % of the samples of one instantiation, deterministic)
    differed from its working neighbours in `v_accvgpr_write_b32` spill code INSIDE EXEC-narrowed regions (lane-dependent loops of a positional-
    encoding store + 64 more live registers).  The mechanism below the ISA is not established; the shipped UV kernels are kept free of the
    pattern: no v_accvgpr_write between an instruction that narrows EXEC (s_and_saveexec / s_and_b64 exec / v_cmpx / s_mov_b64 exec, <non -1>)
    and the instruction that restores it (s_or_b64 exec / s_mov_b64 exec, -1 / a label that ends the region).
    -> [(kernel, instruction index, line)], number of v_accvgpr_write seen."""
    bad, seen = [], 0
    for name, body in kernels(path):
        if not re.search(pattern, name):
            continue
        narrowed = False
        k = 0
        for l in body:
            p = parse(l)
            if not p:
                continue
            op, ops = p
            k += 1
            joined = ", ".join(ops)
            if op.startswith(("s_and_saveexec", "s_andn2_saveexec", "s_or_saveexec", "v_cmpx")) or (op in ("s_and_b64", "s_andn2_b64", "s_xor_b64") and ops and ops[0] == "exec"):
                narrowed = True
            elif (op in ("s_or_b64",) and ops and ops[0] == "exec") or (op == "s_mov_b64" and ops and ops[0] == "exec" and ops[1].strip() == "-1"):
                narrowed = False
            elif op == "s_mov_b64" and ops and ops[0] == "exec":
                narrowed = True
            if op.startswith("v_accvgpr_write"):
                seen += 1
                if narrowed:
                    bad.append((name, k, op + " " + joined))
    return bad, seen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernels", default=".*")
    ap.add_argument("--window", type=int, default=64)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--forms", action="store_true", help="print the packed fp32 instruction forms of the selected kernels")
    ap.add_argument("--lint", action="store_true", help="exit 1 if a kernel with bf16 matrix instructions holds a forbidden packed form")
    ap.add_argument("--hand", action="store_true", help="check the hand-placed wait states: matrix write -> AGPR-data LDS store, VALU write -> DPP read, "
                    "v_accvgpr_write inside EXEC-narrowed regions (exit 1 on a hit)")
    a = ap.parse_args()
    if a.forms:
        for name, body in kernels(a.asm):
            if re.search(a.kernels, name):
                print(name[:150])
                for (op, mods), n in sorted(pk_forms(body).items()):
                    print(f"    {n:5d}  {op} {mods}" + ("      <-- forbidden in bf16-MFMA kernels" if forbidden((op, mods)) else ""))
        return 0
    if a.hand:
        b1, n1 = lint_mfma_to_lds(a.asm, a.kernels)
        b2, n2 = lint_valu_to_dpp(a.asm, a.kernels)
        b3, n3 = lint_accvgpr_write_in_exec_regions(a.asm, a.kernels)
        for name, i, st, wr, have, req in b1[:20]:
            print(f"{name[:100]} @{i}: {st}  <- {wr}: {have} wait states, {req} required")
        for name, i, dpp, wr, have in b2[:20]:
            print(f"{name[:100]} @{i}: {dpp}  <- {wr}: {have} wait states, 2 required")
        for name, i, line in b3[:20]:
            print(f"{name[:100]} @{i}: {line} inside an EXEC-narrowed region")
        print(f"{n1} AGPR-data LDS stores checked, {len(b1)} too close to their matrix instruction; {n2} DPP instructions checked, {len(b2)} too close to a VALU write; "
              f"{n3} v_accvgpr_write seen, {len(b3)} inside EXEC-narrowed regions")
        return 1 if (b1 or b2 or b3) else 0
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

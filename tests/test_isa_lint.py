"""Static fence around round 2's timing-dependent InfoInv NGF_F_SPLIT_BF16 pass: hipcc's packed fp32 code for the
positional-factor chain (v_pk_mov_b32, crossed-halves v_pk_mul_f32 / negated v_pk_add_f32) gave accumulators that depended on timing; the same
operations as single VALU instructions are bit-stable.  The mechanism below the ISA is unknown, so the instruction FORMS that only the unstable
build contains (profiles/r04_isa_packed_diff.txt) are forbidden mechanically in every kernel that issues bf16 matrix instructions:
profiles/isa_hazards.py --lint on the assembly of the shipped translation units -- and, so that the fence is known to catch what it is for, on
the -DNGF_EXP_PACKED_PE build, which must be flagged.  Compiles three translation units with -save-temps (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neural-gauge-fields_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "profiles"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function", "-save-temps", "-c"]


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    jobs = {}
    for tag, src, defs in (("field", "ngf_field.hip", []), ("uv", "ngf_uv.hip", []), ("field_packed", "ngf_field.hip", ["-DNGF_EXP_PACKED_PE=1"]),
                           ("uv_short_nops", "ngf_uv.hip", ["-DNGF_EXP_UV_SHORT_NOPS=1"])):
        d = tmp_path_factory.mktemp(tag)
        jobs[tag] = (d, src, subprocess.Popen([hipcc] + FLAGS + defs + [os.path.join(CSRC, src), "-o", "out.o"], cwd=d, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    out = {}
    for tag, (d, src, p) in jobs.items():
        log, _ = p.communicate(timeout=900)
        assert p.returncode == 0, log.decode()[-2000:]
        out[tag] = os.path.join(d, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        assert os.path.exists(out[tag])
    return out


def test_shipped_bf16_kernels_hold_no_forbidden_packed_form(asm):
    import isa_hazards
    for tag in ("field", "uv"):
        names = [n for n, body in isa_hazards.kernels(asm[tag]) if any("v_mfma" in l and "bf16" in l for l in body)]
        assert names, f"{tag}: no kernel with bf16 matrix instructions found -- the lint would be vacuous"
        hits = isa_hazards.lint(asm[tag])
        assert not hits, hits


def test_the_lint_flags_the_unstable_build(asm):
    import isa_hazards
    hits = isa_hazards.lint(asm["field_packed"], "render_kernel.*InfoInvPolicyTILb1")
    forms = {f for _, f, _ in hits}
    assert any(op == "v_pk_mov_b32" for op, _ in forms) and any(op == "v_pk_mul_f32" for op, _ in forms) and any(op == "v_pk_add_f32" for op, _ in forms), hits


def _kernel_blocks(path):
    """{kernel symbol: (body lines, scratch bytes per lane)} of a -save-temps assembly file."""
    import re
    text = open(path).read()
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        out[m.group(1)] = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2)).group(1))
    return out


def test_shipped_hot_kernels_use_no_scratch(asm):
    """DESIGN.md quotes 0 B of scratch for the UV-Mapping kernels (round 4: the split kernel's 140 B were per-layer addresses hipcc had hoisted out of
    the ray loop and spilled) and for the production render kernels of the default levels; held here so a source change that brings spills back fails
    on the CPU."""
    scratch = {**_kernel_blocks(asm["uv"]), **_kernel_blocks(asm["field"])}
    want = ["uv_render_kernelILi2ELb0", "uv_render_kernelILi2ELb1", "uv_render_kernelILi1ELb0",
            "render_kernelINS_14TriPlanePolicyILb1ELb1ELi12ELi1ELb0EEELb1ELb0E",      # level 3 (module default), production instantiation
            "render_kernelINS_14TriPlanePolicyILb1ELb0ELi12ELi1ELb0EEELb1ELb0E",      # level 2
            "render_kernelINS_8MaskSkipINS_14TriPlanePolicyILb1ELb1ELi12ELi1ELb0EEEEELb1ELb0E",      # level 3 for fields with an alpha mask (empty-space skipping, round 6)
            "render_kernelINS_8MaskSkipINS_14TriPlanePolicyILb1ELb0ELi12ELi1ELb0EEEEELb1ELb0E",      # level 2, the same
            "render_kernelINS_8MaskSkipINS_23TriPlaneBakedBf16PolicyEEELb1ELb0E",                     # level 3 + bf16 layer 2, the same
            "render_kernelINS_8MaskSkipINS_14InfoInvPolicyTILb1ELb0EEEEELb1ELb0E",                    # InfoInv split bf16, the same (the fp32 one carries 52 B there: profiles/r06_mask_skip.txt)
            "render_kernelINS_14InfoInvPolicyTILb0ELb0EEELb1ELb0E",                   # InfoInv fp32
            "render_kernelINS_14InfoInvPolicyTILb1ELb0EEELb1ELb0E",                   # InfoInv split bf16
            # round 6: the kernels of a training step -- train_density_bwd_kernel<true, true> carried 28 B per lane through round 5
            # (its two profiling counters lived in an array behind a maybe-null pointer)
            "train_density_kernel", "train_scan_kernel", "train_color_fwd16_kernel", "train_composite_bwd_kernel", "train_color_bwd_kernel",
            "train_bin_scatter_kernel", "train_bin_gather_kernel", "xty_all_kernel", "train_density_bwd_kernel", "train_density_finish_kernel",
            "adam_plane_kernel", "adam_dense_all_kernel"]
    for needle in want:
        hits = {k: v for k, v in scratch.items() if needle in k}
        assert hits, f"no kernel matches {needle}"
        assert all(v == 0 for v in hits.values()), hits


def test_uv_weight_loads_are_global_loads(asm):
    """Round 4 pitfall: an opaque POINTER as per-pass weight base loses its address space -- every 16-byte weight load of
    uv_render_kernel became a flat_load, which counts on both memory counters, and the k loop waited vmcnt(0) lgkmcnt(0) once per trip (+9 % time).
    The shipped kernels stream their weights with global_load_dwordx4 (the fp32 k loops, late in round 4, with buffer_load_dwordx4 -- next test; biases,
    output layers, prologues and the split-bf16 fragments stay global loads); the only flat loads left are the 12-byte camera / background rows."""
    import isa_hazards
    for name, body in isa_hazards.kernels(asm["uv"]):
        if "uv_render_kernel" not in name:
            continue
        assert not [l for l in body if "flat_load_dwordx4" in l], name
        assert sum("global_load_dwordx4" in l for l in body) > 100, name


def test_uv_fp32_k_loops_have_no_vector_instruction_between_the_mfmas_of_a_group(asm):
    """DESIGN.md 4.2 (round 4, late): fp32 MFMA and VALU share the SIMD's datapath and a lone wave pays ~38 cycles for every matrix -> vector -> matrix
    switch, so the one-wave-per-SIMD k loops of uv_render_kernel<2, false> are [B operands + activations of a group] then 16 x (8 MFMAs, one
    BUFFER load whose k-step offset is an SGPR).  A weight load that needs vector address arithmetic again (a global load: 3 % slower than the burst it
    replaced), or a scheduler that spreads the activations among the MFMAs, shows up here as more vector gaps than one per 128 MFMAs."""
    import re
    import isa_hazards
    checked = 0
    for name, body in isa_hazards.kernels(asm["uv"]):
        if "uv_render_kernelILi2ELb0" not in name:
            continue
        blocks, cur, label = [], [], None          # loop bodies: from a label to the branch back to it
        for l in body:
            t = l.strip()
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                label, cur = m.group(1), []
            elif t and not t.startswith((";", ".")):
                cur.append(t)
                if label and re.match(r"^s_cbranch_\w+ " + re.escape(label) + r"$", t):
                    blocks.append(cur)
                    label, cur = None, []
        for b in blocks:
            idx = [i for i, t in enumerate(b) if t.startswith("v_mfma_f32_16x16x4_f32")]
            if len(idx) < 256:
                continue
            gaps = [b[i + 1:j] for i, j in zip(idx, idx[1:])]
            vector_gaps = sum(any(t.startswith("v_") for t in g) for g in gaps)
            loads_between = sum(sum(t.startswith("buffer_load_dwordx4") for t in g) for g in gaps)
            assert vector_gaps <= len(idx) // 128, (name, len(idx), vector_gaps)
            assert loads_between >= len(idx) // 8 - 8, (name, len(idx), loads_between)
            assert not any(t.startswith("global_load") for g in gaps for t in g if len(g) < 4), name
            checked += 1
    assert checked >= 4, checked          # the geometry / texture hidden runs and block2.0, at least


def test_infoinv_density_pass_keeps_its_matrix_instructions_in_runs(asm):
    """DESIGN.md 9 (round 4, late): the fp32 InfoInv density pass issues a plane's twelve operand exchanges, then its 24 v_mfma_f32_32x32x2_f32 in a row, and
    layer 2's 32 ReLUs, then its 32 MFMAs -- written `swap, 2 MFMAs, swap ...` it alternated between the vector and the matrix pipe 52 times per
    64-sample pass.  Pure instructions are not ordered by sched_barrier (the instruction selector had emitted half the ReLUs among the last MFMAs of layer 1
    until the accumulators went through an opaque asm), so the property is held here on the shipped ISA."""
    import re
    import isa_hazards
    checked = 0
    for name, body in isa_hazards.kernels(asm["field"]):
        if "render_kernelINS_14InfoInvPolicyTILb0ELb0EEELb1ELb0E" not in name:
            continue
        blocks, cur = [], []
        for l in body:
            t = l.strip()
            if re.match(r"^\.LBB\d+_\d+:", t):
                blocks.append(cur)
                cur = []
            elif t and not t.startswith((";", ".")):
                cur.append(t)
        blocks.append(cur)
        for b in blocks:
            kinds = ["M" if t.startswith("v_mfma_f32_32x32x2") else "V" for t in b if t.startswith("v_")]
            if "M" not in kinds:
                continue
            runs = sum(1 for i, k in enumerate(kinds) if k == "M" and (i == 0 or kinds[i - 1] != "M"))
            assert runs <= 2, (name, runs, kinds.count("M"))
            checked += kinds.count("M")
    assert checked == 104, checked          # 3 planes x 24 + 32


# ---- hand-placed wait states (round 5) ------------------------------------------------------------------------------
def test_matrix_results_stored_to_lds_from_agprs_wait_long_enough(asm):
    """csrc/ngf_uv.hpp stores a layer's rows with `ds_write2st64_b32 v, a, a` written in inline assembly (LDS stores take AGPR data operands;
    from C++ hipcc copies every accumulator to a VGPR first) -- a matrix-write -> memory-read hazard hipcc's recogniser does not see.  The
    shipped assembly must keep every such store >= 12 (16x16 shapes) / 20 (32x32) wait states behind the matrix instruction that writes its
    data register; the -DNGF_EXP_UV_SHORT_NOPS build (the hand-counted `s_nop 15; s_nop 3` cut to `s_nop 1`) must be flagged."""
    import isa_hazards
    bad, checked = isa_hazards.lint_mfma_to_lds(asm["uv"], "uv_render_kernel")
    assert checked >= 900, checked                      # 744 + 248 stores in the two fp32 kernels
    assert not bad, bad[:5]
    bad_short, checked_short = isa_hazards.lint_mfma_to_lds(asm["uv_short_nops"], "uv_render_kernel")
    assert checked_short == checked and bad_short, "the lint does not see the shortened wait states"
    # no other translation unit stores from AGPRs by hand
    assert isa_hazards.lint_mfma_to_lds(asm["field"])[1] == 0


def test_dpp_reads_keep_two_wait_states_behind_valu_writes(asm):
    """The transmittance / acc / depth chains of the split march are in-place DPP instructions in inline assembly (csrc/ngf_render.hpp
    split_chain, split_chain_rows; the trainer's scan): a VALU write of a VGPR followed by a DPP read of it needs 2 wait states.  Held on the
    shipped assembly -- and shown to bite on a synthetic kernel."""
    import isa_hazards
    bad, checked = isa_hazards.lint_valu_to_dpp(asm["field"])
    assert checked > 5000, checked
    assert not bad, bad[:5]
    import tempfile
    snippet = "_Zfake:\n\tv_add_f32_e32 v1, v2, v3\n\ts_nop 0\n\tv_add_f32_dpp v4, v1, v1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_endpgm\n"
    ok = snippet.replace("s_nop 0", "s_nop 1")
    with tempfile.TemporaryDirectory() as d:
        for text, want in ((snippet, 1), (ok, 0)):
            path = os.path.join(d, "k.s")
            open(path, "w").write(text)
            got, n = isa_hazards.lint_valu_to_dpp(path)
            assert n == 1 and len(got) == want, (text, got)


def test_uv_kernels_hold_no_agpr_copy_inside_exec_narrowed_regions(asm):
    """the one intermediate build of uv_render_kernel with wrong densities differed from its working neighbours in AGPR spill
    copies (v_accvgpr_write_b32) inside EXEC-narrowed regions; the cause below the source is not established, the shipped kernels are kept free of the
    pattern (no divergent control flow between ray set-up and compositing).  The lint's region tracking is checked on a synthetic kernel."""
    import isa_hazards
    bad, seen = isa_hazards.lint_accvgpr_write_in_exec_regions(asm["uv"], "uv_render_kernel")
    assert not bad, bad[:5]
    import tempfile
    text = ("_Zfake:\n\tv_accvgpr_write_b32 a0, v1\n\ts_and_saveexec_b64 s[0:1], vcc\n\tv_accvgpr_write_b32 a1, v2\n\ts_or_b64 exec, exec, s[0:1]\n"
            "\tv_accvgpr_write_b32 a2, v3\n\ts_endpgm\n")
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "k.s")
        open(path, "w").write(text)
        got, n = isa_hazards.lint_accvgpr_write_in_exec_regions(path)
        assert n == 3 and len(got) == 1 and "a1" in got[0][2], got

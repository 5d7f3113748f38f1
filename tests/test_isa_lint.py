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
    for tag, src, defs in (("field", "ngf_field.hip", []), ("uv", "ngf_uv.hip", []), ("field_packed", "ngf_field.hip", ["-DNGF_EXP_PACKED_PE=1"])):
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

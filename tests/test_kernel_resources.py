"""CPU guard on the register budgets the measured numbers rest on (no GPU: reads the gfx950 code objects hipcc produced).

The kernels below were tuned to a residency; a source or compiler change that silently pushes one of them over its budget
(spills, one wave less per SIMD) would cost tens of percent on the GPU and nothing would fail.  The limits are the ones
DESIGN.md section 4 states."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(obj, tmp):
    """{mangled name: {field: int}} of the gfx950 code object embedded in an object file"""
    fat = os.path.join(tmp, "x.fatbin")
    co = os.path.join(tmp, "x.co")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj, os.devnull])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
    notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
    out = {}
    for block in re.split(r"\n  - \.", notes):          # kernel entries sit at indentation 2 (their argument lists deeper)
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        out[name.group(1)] = {k: int(v) for k, v in re.findall(r"\.?(\w+):\s+(\d+)\s*$", block, flags=re.M)}
    return out    # (.vgpr_count is the unified total on gfx950: architected VGPRs + AGPRs)


@pytest.fixture(scope="module")
def objdir():
    from permuto_sdf_amd import build
    build.build(verbose=False)
    for tool in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM, tool)):
            pytest.skip("ROCm LLVM tools not found")
    return build.OBJDIR


def _one(kernels, pattern):
    hits = [v for k, v in kernels.items() if re.search(pattern, k)]
    assert len(hits) == 1, (pattern, [k for k in kernels if re.search(pattern, k)])
    return hits[0]


def test_encode_kernels_keep_their_residency(objdir, tmp_path):
    k = _kernels(os.path.join(objdir, "encode.o"), str(tmp_path))
    # the forward: <P, F, PLAIN>.  PLAIN (no skip mask, no touched-block map: the bench's and the hot path's forward) must fit
    # 8 waves per SIMD in BOTH register files -- 800 scalar registers per SIMD: at most 96 (+ VCC etc.: 102 allocated) per wave;
    # the general instantiation is allowed 7 (it carries five more arguments)
    fwd = _one(k, r"encode_fwd_kernelILi3ELi2ELb1E")
    assert fwd["vgpr_count"] <= 64 and fwd["vgpr_spill_count"] == 0 and fwd["private_segment_fixed_size"] == 0
    assert fwd["sgpr_count"] <= 102 and fwd["sgpr_spill_count"] == 0, fwd
    gen = _one(k, r"encode_fwd_kernelILi3ELi2ELb0E")
    assert gen["vgpr_count"] <= 64 and gen["vgpr_spill_count"] == 0 and gen["sgpr_spill_count"] == 0, gen
    # queue-mode binning kernels of the SDF lattice (pos_dim 3, 2 features): 5 waves per SIMD = at most 96 registers, no spill
    # (template arguments: P, F, LATTICE, POS, QUEUE, DBL -- the last one = the double backward's scatter through the same kernel)
    for pat in (r"encode_bwd_kernelILi3ELi2ELb1ELb0ELb1ELb0E", r"encode_bwd_kernelILi3ELi2ELb1ELb1ELb1ELb0E",
                r"encode_bwd_kernelILi3ELi2ELb1ELb0ELb1ELb1E"):
        b = _one(k, pat)
        assert b["vgpr_count"] <= 96, b
        assert b["vgpr_spill_count"] == 0, b
    # the other instantiations ask for 4 waves (128 registers)
    for pat in (r"encode_bwd_kernelILi4ELi2ELb1ELb0ELb1ELb0E", r"encode_bwd_kernelILi3ELi4ELb1ELb0ELb1ELb0E",
                r"encode_bwd_kernelILi4ELi2ELb1ELb0ELb1ELb1E", r"encode_bwd_kernelILi3ELi4ELb1ELb0ELb1ELb1E"):
        b = _one(k, pat)
        assert b["vgpr_count"] <= 128 and b["vgpr_spill_count"] == 0, b
    red = _one(k, r"encode_bwd_reduce_kernelILi2E")
    assert red["vgpr_count"] <= 64 and red["vgpr_spill_count"] == 0


def test_split_bf16_backward_of_the_baseline_net_does_not_spill(objdir, tmp_path):
    """One wave per SIMD by design (176 persistent accumulators): the whole 512-register file, but nothing in scratch."""
    k = _kernels(os.path.join(objdir, "mlp_bwd_split_double.o"), str(tmp_path))
    b = _one(k, r"mlp_bwd_split_kernelILi3ELb1E")
    assert b["vgpr_count"] <= 512 and b["agpr_count"] >= 176        # the persistent dW accumulators live in AGPRs
    assert b["vgpr_spill_count"] == 0 and b["private_segment_fixed_size"] == 0, b


def test_split_bf16_forward_fits_two_workgroups_per_cu(objdir, tmp_path):
    k = _kernels(os.path.join(objdir, "mlp.o"), str(tmp_path))
    f = _one(k, r"mlp_fwd_split_kernelILi2ELi2ELi2ELi1ELb1ELi3ELb0E")       # 36 -> 64 x 3 -> 1, the bench net, bf16 pieces
    assert f["vgpr_count"] <= 128 and f["vgpr_spill_count"] == 0, f
    f16 = _one(k, r"mlp_fwd_split_kernelILi2ELi2ELi2ELi1ELb1ELi3ELb1E")     # the same net, two fp16 pieces (round 3)
    assert f16["vgpr_count"] <= 256 and f16["vgpr_spill_count"] == 0 and f16["private_segment_fixed_size"] == 0, f16


def test_round3_kernels_do_not_spill(objdir, tmp_path):
    """the fp16 two-piece backward of the BASELINE net (one wave per SIMD, accumulators in AGPRs), the workgroup-cooperative
    backward of the background density net (52 -> 64 x 3 -> 65: its single-wave predecessor spilled 213-227 registers) and the
    fused compositing kernels"""
    k = _kernels(os.path.join(objdir, "mlp_bwd_split_f16.o"), str(tmp_path))
    # template argument: input tiles (3: K0 <= 48, gelu' of the inner layers in LDS; 4: K0 <= 64).  The 176 (192) dW accumulators
    # are pinned in AGPRs; in the instantiation the bench runs nothing else lives there (round 4: it was 240 AGPRs with
    # 100 - 230 v_accvgpr moves per tile).  Exactly these two instantiations exist.
    assert sorted(n for n in k if "mlp_bwd_split_f16_kernel" in n) == sorted(
        n for n in k if re.search(r"mlp_bwd_split_f16_kernelILi[34]EEEv", n)) and sum("mlp_bwd_split_f16_kernel" in n for n in k) == 2
    for pat, acc in ((r"mlp_bwd_split_f16_kernelILi3EEEv", 176), (r"mlp_bwd_split_f16_kernelILi4EEEv", 192)):
        b = _one(k, pat)
        assert b["vgpr_count"] <= 512 and b["agpr_count"] >= acc, b
        assert b["vgpr_spill_count"] == 0 and b["private_segment_fixed_size"] == 0, b
    assert _one(k, r"mlp_bwd_split_f16_kernelILi3EEEv")["agpr_count"] <= 184
    # (the two two-waves-per-SIMD forms of round 5 left the library in round 6: attic/rejected/)
    assert not [n for n in k if "pair_kernel" in n or "cd_kernel" in n]
    k = _kernels(os.path.join(objdir, "mlp_wide.o"), str(tmp_path))
    for pat in (r"mlp_wide_bwd_kernelILi7ELi8ELi8ELi4ELi1E", r"mlp_wide_bwd_kernelILi4ELi4ELi4ELi4ELi5E"):
        b = _one(k, pat)
        assert b["vgpr_count"] <= 256 and b["vgpr_spill_count"] == 0 and b["private_segment_fixed_size"] == 0, b   # 2 waves / SIMD
    # round 6, the same on the fp16 matrix pipe (colour network, background density net, background colour head = two hidden
    # layers): none spills.  (Until the record addresses of the weight prefetch were recomputed per tile instead of hoisted, the
    # colour network's instantiation parked 33 - 72 registers in scratch and reloaded eight address pairs INSIDE the tile loop,
    # each reload waiting for the LDS-DMA prefetch in flight: 108 -> 91 us at a training step's 49 K samples.)
    for pat in (r"mlp_wide_bwd_f16_kernelILi7ELi8ELi8ELi4ELi1E", r"mlp_wide_bwd_f16_kernelILi4ELi4ELi4ELi4ELi5E",
                r"mlp_wide_bwd_f16_kernelILi5ELi4ELi4ELi0ELi1E"):
        b = _one(k, pat)
        assert b["vgpr_count"] <= 256 and b["vgpr_spill_count"] == 0 and b["private_segment_fixed_size"] == 0, b
    k = _kernels(os.path.join(objdir, "composite_fused.o"), str(tmp_path))
    for pat in (r"neus_composite_fwd_kernel", r"neus_composite_bwd_kernelILi2E", r"neus_composite_bwd_kernelILi4E"):
        b = _one(k, pat)
        assert b["vgpr_count"] <= 64 and b["vgpr_spill_count"] == 0, b


# Every kernel of the library that spills registers, with the route a call takes to reach it.  The list is CLOSED: a source or
# compiler change that makes another kernel spill -- or brings back an instantiation nothing dispatches to -- fails here.
# (Round 4 removed the single-wave dW instantiations of the 64-wide nets with many outputs, 128 - 253 spilled registers: their
# parameter gradients come from the workgroup-cooperative kernel of mlp_wide.hip, psdf_mlp_backward returns -2 where that one
# cannot run.)
SPILLING = {
    # BASELINE net 64x3 -> 1 with parameter gradients BELOW 2^18 samples (psdf_mlp_backward: the split-operand kernels take
    # over from 2^18 on; smoke()'s 3 072-sample case and every small-batch test run these), dX requested or not
    r"mlp_bwd_kernelILi3ELi4ELi4ELi4ELi1ELb1ELb1ELb1ELi4E": 64, r"mlp_bwd_kernelILi3ELi4ELi4ELi4ELi1ELb1ELb0ELb1ELi4E": 64,
    r"mlp_bwd_kernelILi4ELi4ELi4ELi4ELi1ELb1ELb1ELb1ELi4E": 128, r"mlp_bwd_kernelILi4ELi4ELi4ELi4ELi1ELb1ELb0ELb1ELi4E": 80,
    r"mlp_bwd_kernelILi2ELi4ELi4ELi4ELi1ELb1ELb1ELb1ELi4E": 32,
    # the reference's SDF net 52 -> 32x3 -> 33 with parameter gradients, two waves per SIMD at the 256-register limit: batches
    # of 2^17 samples and more only (round 5: a training step's ~49 K samples run the one-wave-per-SIMD instantiation, which does
    # not spill and is faster there -- launch_bwd in csrc/mlp_bwd.hip)
    r"mlp_bwd_kernelILi4ELi2ELi2ELi2ELi3ELb0ELb1ELb1ELi8E": 24, r"mlp_bwd_kernelILi4ELi2ELi2ELi2ELi3ELb0ELb0ELb1ELi8E": 8,
    # double backward of the BASELINE net (psdf_mlp_double_backward; the reference's own net is 32 wide and does not spill):
    # fp32-MFMA form, one wave per SIMD -- DESIGN.md "Next": the workgroup-cooperative split form
    r"mlp_dbl_bwd_kernelILi3ELi4ELi4ELi4ELi1ELb1ELb0EE": 256, r"mlp_dbl_bwd_kernelILi4ELi4ELi4ELi4ELi1ELb1ELb0EE": 320,
    # background colour head 80 -> 64x2 -> 3 with parameter gradients (models.py:463-469): every training step, one register
    r"mlp_bwd_kernelILi5ELi4ELi4ELi0ELi1ELb1ELb1ELb1ELi4E": 8,
    # fused encode -> MLP forward of a 32-wide net with 33 outputs (psdf_encode_mlp_forward: the sphere tracer's colour pass)
    r"fused_fwd_kernelILi2ELi2ELi2ELi2ELb0EE": 8,
}


def test_the_list_of_spilling_kernels_is_closed(objdir, tmp_path):
    seen = {}
    for f in sorted(os.listdir(objdir)):
        if not f.endswith(".o"):
            continue
        d = tmp_path / f[:-2]
        d.mkdir()
        for name, k in _kernels(os.path.join(objdir, f), str(d)).items():
            if k.get("vgpr_spill_count", 0) > 0:
                seen[name] = k["vgpr_spill_count"]
    unlisted = {n: v for n, v in seen.items() if not any(re.search(p, n) for p in SPILLING)}
    assert not unlisted, "kernels that spill without a stated route: %r" % unlisted
    for pat, limit in SPILLING.items():
        hits = {n: v for n, v in seen.items() if re.search(pat, n)}
        assert len(hits) <= 1, (pat, hits)
        for n, v in hits.items():
            assert v <= limit, "%s spills %d registers (listed with at most %d)" % (n, v, limit)
    # the instantiations removed in round 4 stay removed
    k = _kernels(os.path.join(objdir, "mlp_bwd.o"), str(tmp_path))
    assert not [n for n in k if re.search(r"mlp_bwd_kernelILi[34]ELi4ELi4ELi4ELi[35]ELb0ELb[01]ELb1E", n)]

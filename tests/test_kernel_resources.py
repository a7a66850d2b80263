"""Build gate on the SHIPPED code objects: every kernel of libfvp_hip.so has zero SGPR / VGPR
spills and no packed-f32 VALU between MFMAs, read from the library itself (llvm-readelf --notes + llvm-objdump -d via
tools/kernel_resources.py) - not from a side compile.  CPU only: the code objects are cross-compiled by build()."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR  # noqa: E402

LIB = os.path.join(ROOT, "faster-voxelpose_amd", "libfvp_hip.so")

# Recorded exceptions (round 6: the LDS-staged / gather forms of the fused projection moved into the diagnostics build, so the
# shipped library holds k_project_triplane_blk only).  Two remain, capped at the recorded counts so that they cannot grow
# unnoticed; everything else: zero.
ALTERNATE_FORMS = {
    # the UNCACHED 20-channel block form (coordinate cache switched off or above its 2 GB limit - no BASELINE configuration):
    # 4 SGPR spills in the set-up code in front of the loops (the camera table of a view is 24 scalars)
    "k_project_triplane_blk<2, false, false, false>": (4, 0),
    "k_project_triplane_blk<2, false, false, true>": (2, 0),      # its five-quad form (JP = 20, round 6)
    # soft-argmax + WeightNet: 24 SGPR spills in the feature loop.  The three spill-free forms tried in round 5 (template
    # on F without the predicates + scalars re-read behind an opaque pointer, per window / per feature group / chained to an
    # earlier feature's result) measured 252-299 us against 177 us per launch: the spills are the faster code.
    "k_softargmax_weightnet": (24, 0),
}


@pytest.fixture(scope="module")
def rows():
    if not os.path.isfile(LIB):
        pytest.fail("libfvp_hip.so is not built (run __graft_entry__.build())")
    if not os.path.isfile(os.path.join(KR.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    rows = KR.scan_library(LIB)
    names = KR.demangle([r["name"] for r in rows])
    for r, d in zip(rows, names):
        r["demangled"] = d
    return rows


def _alternate(r):
    for key, cap in ALTERNATE_FORMS.items():
        if "fvp::" + key in r["demangled"]:
            return cap
    return None


def test_library_holds_the_expected_kernels(rows):
    names = " ".join(r["demangled"] for r in rows)
    for k in ("k_project_triplane_lds", "k_project_triplane<"):      # diagnostics-build code (round 6)
        assert k not in names, k
    assert len(rows) >= 150
    for k in ("k_conv_wino<2, 4, 8, 2, true, false, 2>", "k_conv_reg<128, 4, 1, true>", "k_conv_dma<7, 7, 1, 4, true",
              "k_project_triplane_blk<1, true, false, false>", "k_project_triplane_blk<2, true, false, true>", "k_project_whole_q", "k_conv1d_fused", "k_softargmax_weightnet",
              "k_bb_conv_dma", "k_conv7<64, 4, 4>", "k_conv7<64, 5, 4>", "k_conv7<128, 4, 4>"):
        assert k in names, k


def test_no_shipped_kernel_spills(rows):
    bad = []
    for r in rows:
        cap = _alternate(r)
        if cap is None:
            if r["sgpr_spill"] or r["vgpr_spill"] or r["scratch"]:
                bad.append((r["demangled"].split("(")[0], r["sgpr_spill"], r["vgpr_spill"], r["scratch"]))
        elif r["sgpr_spill"] > cap[0] or r["vgpr_spill"] > cap[1]:
            bad.append((r["demangled"].split("(")[0], r["sgpr_spill"], r["vgpr_spill"], "over the recorded cap"))
    assert not bad, "kernels with register spills: %s" % bad


def test_no_packed_f32_valu_between_mfmas(rows):
    """MI355X_MICROARCH.md: v_pk_{add,mul,fma}_f32 beside fp32 MFMAs cost +22..26 cycles per pair against scalar VALU ("an
    anti-lever, including when the compiler SLP-packs adjacent scalar f32 adds"); the Winograd TU is built with
    -fno-slp-vectorize for that reason."""
    bad = [(r["demangled"].split("(")[0], r["packed_f32_between_mfma"]) for r in rows if r["packed_f32_between_mfma"]]
    assert not bad, bad


def test_winograd_kernels_fit_two_waves_per_simd(rows):
    w = [r for r in rows if "k_conv_wino<" in r["demangled"]]
    assert len(w) == 72                               # 56 two-block instances + 16 quarter-size (one block per wave) ones
    for r in w:
        assert r["vgpr"] + r["agpr"] <= 256, r
        args = r["demangled"].split("k_conv_wino<")[1].split(">")[0].split(",")
        cc, cw = int(args[2]), int(args[6])
        assert r["mfma"] == 8 * cc * cw, r      # two chunk bodies (first / other) x CC/4 steps x 16 CW MFMAs: straight-line


def test_front_conv7_is_straight_line_and_fits_three_waves_per_simd(rows):
    """k_conv7: NCG x 196 MFMAs of fully unrolled code per tile; the four-group instances (15 joints) within 168 registers."""
    k7 = [r for r in rows if "k_conv7<" in r["demangled"]]
    assert len(k7) == 4
    for r in k7:
        args = r["demangled"].split("k_conv7<")[1].split(">")[0].split(",")
        ncg = int(args[1])
        assert r["mfma"] == 196 * ncg, r
        if ncg == 4:
            assert r["vgpr"] + r["agpr"] <= 168, r


def test_no_readfirstlane_feeds_an_asm_buffer_access(rows):
    """the Winograd epilogue re-points an SGPR descriptor and issues inline-asm buffer accesses right behind
    it; that is safe while the new base is produced by SALU.  If it ever became a v_readfirstlane result, the
    VALU-writes-SGPR -> VMEM hazard (5 wait states) would be violated silently: the gate counts such pairs in the ISA."""
    bad = [(r["demangled"].split("(")[0], r["readfirstlane_to_buffer_hazards"]) for r in rows if r["readfirstlane_to_buffer_hazards"]]
    assert not bad, bad


def test_backbone_kernels_are_gated_too(rows):
    """Round 6: the bf16 backbone's code objects under the same gate - no spills anywhere (covered
    by test_no_shipped_kernel_spills: no exception entry names a k_bb_ kernel), and the fused kernels keep the shape their
    occupancy plan relies on: k_bb_stem_pool <= 102 registers (two 9-wave workgroups per CU: five waves on a SIMD) with its 28
    MFMAs per conv row unrolled; k_bb_bottleneck64 inside the 256 registers of two waves per SIMD with the 36 W2 operands
    (144 registers) resident, i.e. no scratch."""
    bb = [r for r in rows if "fvp::k_bb_" in r["demangled"]]
    names = " ".join(r["demangled"] for r in bb)
    for k in ("k_bb_stem_pool", "k_bb_bottleneck64<64, true>", "k_bb_bottleneck64<256, false>", "k_bb_conv_dma<256, 64, 2, false>",
              "k_bb_conv_dma<256, 64, 2, true>", "k_bb_conv<64>", "k_bb_conv<128>"):
        assert k in names, k
    assert not any(k.startswith("k_bb") for k in ALTERNATE_FORMS)
    for r in bb:
        assert r["sgpr_spill"] == 0 and r["vgpr_spill"] == 0 and r["scratch"] == 0, r
        if "k_bb_stem_pool" in r["demangled"]:
            assert r["vgpr"] + r["agpr"] <= 102 and r["mfma"] == 28, r
        if "k_bb_bottleneck64" in r["demangled"]:
            assert 144 < r["vgpr"] + r["agpr"] <= 256, r
            # conv1: 8 MFMAs per 64-channel block, two loop bodies when there is more than one block; conv2: 72; conv3 (+ downsample): 4 (+ 4)
            assert r["mfma"] >= 72 + 8 + 4, r

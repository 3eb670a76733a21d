"""The systolic kernels are designed around 4 waves per SIMD (one workgroup of every alignment of a
1024-read batch resident at once): that holds only while each stays within 128 VGPRs and uses no
scratch.  The register allocator has crossed that line silently before (an inlined helper with its
own constants, a loop the optimiser decided to clone), halving the measured throughput -- so the
budget is checked from the compiler's own metadata.  CPU-only: hipcc cross-compiles gfx950."""
import os
import re
import shutil
import subprocess

import pytest

from cpecan_load import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("rows,suffix", [(4, ""), (3, "_r3"), (2, "_r2"), (1, "_r1")])
def test_systolic_kernels_fit_four_waves_per_simd(tmp_path, rows, suffix):
    src = os.path.join(ROOT, "cpecan-signal_amd", "csrc", "cpecan_kernel_systolic.hip")
    out = str(tmp_path / "sy.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fno-fast-math", "-Wno-unused-function", "-DSY_R=%d" % rows,
                           "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", out, src],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    for name in ("cpecan_k_sy_forward" + suffix, "cpecan_k_sy_backward" + suffix):
        meta = text[text.index(".name:           " + name):]
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
        body = text[text.index("\n" + name + ":"):]
        body = body[:body.index("s_endpgm")]
        assert vgpr <= 128, "%s uses %d VGPRs: fewer than 4 waves per SIMD" % (name, vgpr)
        assert spill == 0 and "scratch_" not in body, "%s spills to scratch" % name


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("cells,hdp", [(3, False), (2, False), (3, True), (2, "vanilla")])
def test_wave_kernels_fit_two_waves_per_simd(tmp_path, cells, hdp):
    """The wave-per-alignment sweeps run as one forward and one backward wave per SIMD (the forward sweep of window
    w+1 beside the backward sweep of window w): the two allocations (unified register file of 512 per lane, handed
    out in blocks of 8) must fit together, with nothing in scratch.  The four-cell build is the documented exception
    (its backward sweep takes 306 registers: one wave per SIMD while it runs)."""
    src = os.path.join(ROOT, "cpecan-signal_amd", "csrc", "cpecan_kernel_wave.hip")
    out = str(tmp_path / "wv.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fno-fast-math", "-Wno-unused-function", "-DWV_L=%d" % cells] +
                          (["-DWV_VANILLA"] if hdp == "vanilla" else ["-DWV_HDP"] if hdp else []) +
                          ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-S",
                           "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    text = open(out).read()
    sfx = "_%s%d" % ("v" if hdp == "vanilla" else "h" if hdp else "l", cells)
    alloc = {}
    for name in ("cpecan_k_wv_forward" + sfx, "cpecan_k_wv_backward" + sfx, "cpecan_k_wv_resweep" + sfx,
                 "cpecan_k_wv_backward_em" + sfx):  # (every machine's E-step sweeps back with this one)
        meta = text[text.index(".name:           " + name + "\n"):]
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
        alloc[name] = (vgpr + 7) // 8 * 8
        assert spill == 0, "%s spills %d VGPRs to scratch" % (name, spill)
    fwd = alloc.pop("cpecan_k_wv_forward" + sfx)
    for name, a in alloc.items():
        assert fwd + a <= 512, "%s (%d registers) and the forward sweep (%d) no longer share a SIMD" % (name, a, fwd)
    if hdp is True:  # three register pairs per slot instead of ten: the forward sweep is the light one
        meta = text[text.index(".name:           cpecan_k_wv_forward" + sfx + "\n"):]
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= 192


def test_wave5_pair_kernels_fit_two_waves_per_simd(tmp_path):
    """The 5-state machine's two-waves-per-alignment kernels exist to put a forward and a backward wave on every SIMD
    of a small batch: each wave is allocated the registers of its own sweep only (256 at most: two waves per SIMD,
    nothing in scratch); with one and two cells per lane three waves fit (170 registers)."""
    src = os.path.join(ROOT, "cpecan-signal_amd", "csrc", "cpecan_kernel_wave5.hip")
    out = str(tmp_path / "w5.s")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-fno-fast-math", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-S",
                           "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    text = open(out).read()
    for name, cap in (("cpecan_k_wave5p_l1", 170), ("cpecan_k_wave5p_l2", 170), ("cpecan_k_wave5p_l3", 256),
                      ("cpecan_k_wave5pe_l1", 256), ("cpecan_k_wave5pe_l2", 256), ("cpecan_k_wave5pe_l3", 256)):
        meta = text[text.index(".name:           " + name + "\n"):]
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
        assert vgpr <= cap, "%s uses %d VGPRs" % (name, vgpr)
        assert spill == 0, "%s spills %d VGPRs to scratch" % (name, spill)


def test_assembly_sweeps_share_a_simd_and_a_cu():
    """The hand-scheduled sweeps (csrc/asm/cpecan_sweeps_gfx950.s): one forward and one backward wave per SIMD, with
    room left in the register file for the post kernel's waves while the next forward sweep has not begun; four waves of
    each per CU within the 160 KB of LDS; nothing in scratch (the kernels have no scratch segment at all)."""
    text = open(os.path.join(ROOT, "cpecan-signal_amd", "csrc", "asm", "cpecan_sweeps_gfx950.s")).read()
    meta = text[text.index(".amdgpu_metadata"):]
    res = {}
    for name in ("cpecan_k_asm_forward_l3", "cpecan_k_asm_backward_l3"):
        m = meta[meta.index(".name:           " + name + "\n") - 2000: meta.index(".name:           " + name + "\n") + 2000]
        block = meta[:meta.index(".name:           " + name + "\n")].rsplit("  - .agpr_count", 1)[1] + \
            meta[meta.index(".name:           " + name + "\n"):].split("  - .agpr_count", 1)[0]
        res[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                     for k in ("vgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count")}
        del m
    f, b = res["cpecan_k_asm_forward_l3"], res["cpecan_k_asm_backward_l3"]
    assert f["vgpr_count"] <= 256 and b["vgpr_count"] <= 256 and f["vgpr_count"] + b["vgpr_count"] <= 512
    assert 512 - b["vgpr_count"] >= 64   # the post kernel (64 registers) beside a backward wave
    assert 4 * (f["group_segment_fixed_size"] + b["group_segment_fixed_size"]) <= 160 * 1024
    for r in (f, b):
        assert r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0


def test_assembly_sweeps_stay_inside_their_register_allocation():
    """No instruction of a hand-scheduled kernel names a VGPR or an SGPR beyond what its descriptor asks for: registers
    past the allocation are another wave's (the CPU emulator of tests/gcn_emu.py has all 256 and would not notice)."""
    text = open(os.path.join(ROOT, "cpecan-signal_amd", "csrc", "asm", "cpecan_sweeps_gfx950.s")).read()
    for name in ("cpecan_k_asm_forward_l3", "cpecan_k_asm_backward_l3"):
        body = text[text.index("\n%s:\n" % name):]
        desc = body[body.index(".amdhsa_kernel %s" % name):]
        body = body[:body.index(".amdhsa_kernel %s" % name)]
        nv = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
        ns = int(re.search(r"\.amdhsa_next_free_sgpr (\d+)", desc).group(1))
        top = {"v": -1, "s": -1}
        for kind, a, b in re.findall(r"\b([vs])\[(\d+):(\d+)\]", body):
            top[kind] = max(top[kind], int(b))
        for kind, a in re.findall(r"\b([vs])(\d+)\b", body):
            top[kind] = max(top[kind], int(a))
        assert 0 <= top["v"] < nv, (name, top, nv)
        assert 0 <= top["s"] < ns, (name, top, ns)

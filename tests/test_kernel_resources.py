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

"""The C-ABI library loads, exports every symbol include/cpecan_hip.h declares, and refuses to
compute without a GPU (there is no CPU fallback in the product)."""
import ctypes
import os
import re

import pytest

from cpecan_load import binding, ROOT

cp = binding()


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "cpecan_hip.h")).read()
    declared = set(re.findall(r"\b(cpecan_[a-z0-9_]+)\s*\(", header))
    assert declared == set(cp.EXPORTS), declared ^ set(cp.EXPORTS)
    lib = ctypes.CDLL(cp.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_no_oracle_in_product():
    # the shipped path must not reach into oracle/: no reference to it in product sources or binding
    prod = os.path.join(ROOT, "cpecan-signal_amd")
    for dirpath, _, files in os.walk(prod):
        for f in files:
            if f.endswith((".hip", ".h", ".c", ".cpp", ".py")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "liborc" not in text and "pyoracle" not in text and "cpecan_oracle" not in text, f
    # ... nor do the timing scripts under tools/ (those that use the checker live under tests/tools/)
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith((".py", ".sh", ".hip")):
            text = open(os.path.join(ROOT, "tools", f)).read()
            assert "liborc" not in text and "pyoracle" not in text and "cpecan_oracle" not in text, f


def test_no_wrong_by_construction_switch_in_the_product_library():
    """timing-study switches that compute wrong results exist only in the side libraries tools/ablate_asm.sh builds
    (-DCPECAN_TIMING_BUILD, CPECAN_ASM_ABLATE at generation): not in libcpecan_hip.so, not in the sources it is built
    from by default, not in the assembly it embeds"""
    so = open(os.path.join(ROOT, "cpecan-signal_amd", "libcpecan_hip.so"), "rb").read()
    for name in (b"CPECAN_TIMING_FORWARD_ONLY", b"CPECAN_TIMING_SERIAL", b"CPECAN_TIMING_NO_POST", b"CPECAN_ASM_ABLATE", b"WV_ABL_"):
        assert name not in so, name
    src = os.path.join(ROOT, "cpecan-signal_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")):
            text = open(os.path.join(src, f)).read()
            assert "WV_ABL_" not in text, f
            for m in ("CPECAN_TIMING_FORWARD_ONLY", "CPECAN_TIMING_SERIAL", "CPECAN_TIMING_NO_POST"):
                for line in text.split(m)[:-1]:  # every use sits under #ifdef CPECAN_TIMING_BUILD
                    assert line.rfind("#ifdef CPECAN_TIMING_BUILD") > line.rfind("#endif"), (f, m)
    make = open(os.path.join(ROOT, "cpecan-signal_amd", "Makefile")).read()
    assert "CPECAN_TIMING_BUILD" not in make and "CPECAN_ASM_ABLATE" not in make
    # the tracked assembly is the unablated one: generating it again without switches gives the same text
    import subprocess
    import sys
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        env = {k: v for k, v in os.environ.items() if not k.startswith("CPECAN_ASM")}
        subprocess.check_call([sys.executable, os.path.join(src, "asm", "gen_sweeps.py"), os.path.join(d, "s.s")],
                              env=env, stderr=subprocess.DEVNULL)
        assert open(os.path.join(d, "s.s")).read() == open(os.path.join(src, "asm", "cpecan_sweeps_gfx950.s")).read()


def test_fails_loudly_without_gpu():
    if cp.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(cp.CpecanError) as ei:
        cp.Context(0)
    assert ei.value.code == cp.ENODEVICE
    assert "no CPU path" in str(ei.value)


def test_em_library_exports_declared_symbols():
    """include/cpecan_em.h: libcpecan_em.so loads without a GPU and exports what the header declares"""
    import ctypes as C
    import re
    header = open(os.path.join(ROOT, "include", "cpecan_em.h")).read()
    names = set(re.findall(r"\b(cpecan_em_[a-z_]+)\s*\(", re.sub(r"/\*.*?\*/", "", header, flags=re.S)))
    assert names == {"cpecan_em_run", "cpecan_em_last_error", "cpecan_em_comm_create", "cpecan_em_comm_reduce",
                     "cpecan_em_comm_destroy", "cpecan_em_set_rendezvous_nonce", "cpecan_em_rendezvous_exchange",
                     "cpecan_em_rendezvous_done"}
    lib = C.CDLL(os.path.join(ROOT, "cpecan-signal_amd", "libcpecan_em.so"))
    for n in names:
        assert hasattr(lib, n)
    assert lib.cpecan_em_run(None, 1, C.c_double(0.0), None, None, None) != 0  # argument check, no GPU touched

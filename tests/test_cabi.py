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

import os
import sys

import pytest

try:  # torch brings its own HIP/HSA runtime: it has to be the first one loaded in a process that also uses
    import torch  # noqa: F401  torch.cuda (the EM driver, bench.py); see binding.lib()
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def template_model():
    import pyoracle as o
    match, skip, gapy = o.load_pore_model(os.path.join(GOLDEN, "template_median68pA.model"))
    return match, skip, gapy


@pytest.fixture(scope="session")
def zymo_read():
    import pyoracle as o
    rd = o.load_npread(os.path.join(GOLDEN, "ZymoC_ch_1_file1.npRead"))
    with open(os.path.join(GOLDEN, "ZymoRef.txt")) as f:
        rd["reference"] = f.read().strip()
    return rd

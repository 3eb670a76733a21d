"""CPU-only checks of bench.py's host-side pieces: the all-core CPU baseline (a pool of forked oracle workers)
agrees with single calls, and the command line keeps the contract's flags."""
import os
import subprocess
import sys

import numpy as np

import synth
from harness import band_params, run_oracle_item

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_all_core_cpu_baseline_counts_the_same_cells():
    batch = synth.make_batch(3, 6, 200, 410, anchor_every=50)
    out = bench.cpu_all_cores(batch, 100, 3, 2)
    cells = sum(run_oracle_item(batch, i, band_params(0.01, 1000, 40, 100), (1, 1))["cells"] for i in range(6))
    assert out["cores"] == 2 and out["kind"] == "port" and out["unit"] == "Gcells/s"
    assert out["cells"] == cells and out["value"] > 0


def test_command_line_contract():
    h = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert h.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in h.stdout

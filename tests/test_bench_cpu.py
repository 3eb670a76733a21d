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


def test_gpus_2_starts_its_own_two_ranks():
    """`python bench.py --gpus 2` from a plain command line: the process starts two workers itself (one per rank,
    torch.distributed.run on 127.0.0.1); --rehearse keeps them off the GPU (gloo, host band geometry as the step), so
    the plumbing -- rendezvous, shard by rank, barrier, max-over-ranks time, summed cells, ONE JSON line from rank 0 --
    runs here."""
    import json
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--reads", "6", "--events",
           "400", "--kmers", "200", "--steps", "2", "--warmup", "1", "--master-port", str(29500 + os.getpid() % 2000)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["value"] is None
    assert "REHEARSAL" in out["data"]
    # every rank has its own reads (seeded by rank): the all-rank sum is rank 0's cells plus another rank's
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearse", "--reads", "6",
                          "--events", "400", "--kmers", "200", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=600)
    solo = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert solo["n_gpus"] == 1 and solo["config"]["cells_rank0"] == out["config"]["cells_rank0"]
    assert out["config"]["cells_all_ranks"] > out["config"]["cells_rank0"] > 0
    assert out["config"]["cells_all_ranks"] != 2 * out["config"]["cells_rank0"]


def test_command_line_contract():
    h = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert h.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--mode", "--inflight", "--single-steps"):
        assert flag in h.stdout

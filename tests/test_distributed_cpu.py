"""N > 1 path on the CPU: two gloo ranks shard a small batch, compute Baum-Welch expectations for their
reads (with the oracle standing in for the GPU op) and combine them with one all-reduce; the result must
equal the single-process sum and every rank must derive the same updated model."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import dist_em
import pyoracle as o
import synth
from harness import band_params, run_oracle_item


def _expect_for(batch, idx, bp):
    hmm = o.OrcExpectations()
    for i in idx:
        run_oracle_item(batch, i, bp, (1, 1), expectations=hmm)
    v = np.zeros(dist_em.EXP_LEN)
    v[:9] = hmm.transitions[:]
    v[9:9 + 4096] = hmm.kmerGap[:]
    v[-1] = hmm.likelihood
    return v


def _worker(rank, world, port, out_dir):
    for p in (os.path.dirname(os.path.abspath(__file__)),
              os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = synth.make_batch(41, 6, 120, 250, anchor_every=25, distinct_models=False, length_sigma=0.4)
    bp = band_params(0.01, 80, 20, 40)
    sizes = [it["lX"] + it["lY"] for it in batch["items"]]
    mine = dist_em.shard(sizes, rank, world)
    vec = torch.from_numpy(_expect_for(batch, mine, bp)).reshape(1, -1)
    dist_em.allreduce_expectations(vec, dist)
    trans, gap_x = dist_em.m_step(vec[0].numpy())
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.concatenate([vec[0].numpy(), trans, gap_x]))
    np.save(os.path.join(out_dir, "idx%d.npy" % rank), np.array(mine))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_allreduce(tmp_path):
    world = 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npy")
    r1 = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(r0, r1)  # identical statistics and identical updated model on every rank
    i0 = set(np.load(tmp_path / "idx0.npy").tolist())
    i1 = set(np.load(tmp_path / "idx1.npy").tolist())
    assert i0 | i1 == set(range(6)) and not (i0 & i1)  # every read exactly once
    batch = synth.make_batch(41, 6, 120, 250, anchor_every=25, distinct_models=False, length_sigma=0.4)
    single = _expect_for(batch, range(6), band_params(0.01, 80, 20, 40))
    # summation order differs (per-rank partial sums): 1e-12 relative
    assert np.allclose(r0[:dist_em.EXP_LEN], single, rtol=1e-12, atol=1e-300)
    trans = r0[dist_em.EXP_LEN:dist_em.EXP_LEN + 9]
    assert np.isneginf(trans[8]) and np.all(trans[:7] <= 0)


def _loop_worker(rank, world, port, out_dir):
    for p in (os.path.dirname(os.path.abspath(__file__)),
              os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = synth.make_batch(43, 5, 100, 210, anchor_every=25, distinct_models=False)
    bp = band_params(0.01, 80, 20, 40)
    mine = dist_em.shard([it["lX"] + it["lY"] for it in batch["items"]], rank, world)
    _, gap0, _ = batch["models"][0]

    def e_step(transitions, gap_x):  # the oracle stands in for the GPU op (tests only)
        hmm = o.OrcExpectations()
        saved = batch["models"][0]
        batch["models"][0] = (saved[0], gap_x, saved[2])
        for i in mine:
            run_oracle_item(batch, i, bp, (1, 1), transitions=list(transitions), expectations=hmm)
        batch["models"][0] = saved
        v = np.full(dist_em.EXP_LEN, 1e-9)
        v[:9] += hmm.transitions[:]
        v[9:9 + 4096] += hmm.kmerGap[:]
        v[-1] = hmm.likelihood
        t = torch.from_numpy(v)
        dist_em.allreduce_expectations(t, dist)
        return t.numpy()

    import harness
    lines = []
    r = dist_em.train(e_step, harness.cp.NANOPORE_TRANSITIONS, gap0, 3, log=lines.append)
    np.save(os.path.join(out_dir, "loop%d.npy" % rank),
            np.concatenate([r["transitions"], r["gap_x"], r["running_likelihoods"]]))
    assert len(lines) == 3 and lines[0].startswith("0| ")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_baum_welch_loop(tmp_path):
    """the EM loop driver (cpecan-signal_amd/em.py) on two gloo ranks: the same model and the same running
    likelihoods on every rank, and a likelihood that rises"""
    port = 31500 + os.getpid() % 2000
    mp.spawn(_loop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "loop0.npy"), np.load(tmp_path / "loop1.npy")
    assert np.array_equal(r0, r1)
    like = r0[-3:]
    assert like[0] < like[1] < like[2] < 0
    assert np.isneginf(r0[8]) and np.all(np.isfinite(r0[9:9 + 4096]))  # SWITCH_TO_Y; pseudocounts keep logs finite


def test_shard_is_balanced_and_complete():
    sizes = [100, 900, 300, 300, 50, 700, 10, 400]
    parts = [dist_em.shard(sizes, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == list(range(8))
    loads = [sum(sizes[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(sizes)

"""The Baum-Welch loop driver (cpecan-signal_amd/em.py) on the GPU: its E-step -- one expectation batch, the
per-model blocks summed in HBM through a zero-copy torch view of the C-ABI's device buffer, the all-reduce hook --
against the oracle, and three iterations of the loop (scripts/trainModels.py:244-330 semantics)."""
import numpy as np
import pytest

import dist_em
import pyoracle as o
import synth
from harness import band_params, cp, run_oracle_item

pytestmark = pytest.mark.gpu


def test_gpu_e_step_and_loop():
    ctx = cp.Context(0)
    batch = synth.make_batch(47, 6, 150, 310, anchor_every=30)  # one scaled model per read
    bp = band_params(0.01, 100, 20, 40)
    reads = list(range(len(batch["items"])))
    gap0 = batch["models"][0][1]
    got = dist_em.gpu_e_step(cp, ctx, batch, bp, reads, cp.NANOPORE_TRANSITIONS, gap0)
    hmm = o.OrcExpectations()
    for i in reads:
        run_oracle_item(batch, i, bp, (1, 1), expectations=hmm)
    ref = np.concatenate([np.array(hmm.transitions), np.array(hmm.kmerGap), [hmm.likelihood]])
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-12) and got[-1] < 0

    # two contexts: the reads run as two concurrent batches; same sums up to the order of addition
    ctx2 = cp.Context(0)
    both = dist_em.gpu_e_step(cp, ctx, batch, bp, reads, cp.NANOPORE_TRANSITIONS, gap0, ctx2=ctx2)
    assert np.allclose(both, got, rtol=1e-12, atol=1e-300)
    ctx2.close()

    # the persistent E-step: inputs and emission tables set up once, transitions and gap probabilities rewritten in
    # place per iteration; identical to the one-shot E-step for the same model, also after the model has changed
    ctx3, ctx4 = cp.Context(0), cp.Context(0)
    keep = dist_em.PersistentEStep(cp, [ctx3, ctx4], batch, bp, reads, cp.NANOPORE_TRANSITIONS, gap0)
    assert np.allclose(keep(cp.NANOPORE_TRANSITIONS, gap0), got, rtol=1e-12, atol=1e-300)
    t1, g1 = dist_em.m_step(got + np.r_[np.full(dist_em.EXP_LEN - 1, 1e-9), 0.0])
    again = keep(t1, g1)
    assert np.allclose(again, dist_em.gpu_e_step(cp, ctx, batch, bp, reads, t1, g1), rtol=1e-12, atol=1e-300)
    assert again[-1] > got[-1]  # one EM step raises the likelihood
    keep.close()
    ctx3.close()
    ctx4.close()

    lines = []
    r = dist_em.train(lambda t, g: dist_em.gpu_e_step(cp, ctx, batch, bp, reads, t, g, pseudocount=1e-9),
                      cp.NANOPORE_TRANSITIONS, gap0, 3, log=lines.append)
    like = r["running_likelihoods"]
    assert np.isclose(like[0], ref[-1], rtol=1e-12) and like[0] < like[1] < like[2]
    assert len(lines) == 3 and np.isneginf(r["transitions"][8])
    assert np.allclose(np.exp(r["gap_x"]).sum(), 1.0)
    ctx.close()


def test_native_em_loop_matches_the_python_driver():
    """cpecan_em_run (include/cpecan_em.h, libcpecan_em.so): the same three iterations as a native host loop -- models
    resident, transitions rewritten in place, per-read blocks summed on the device, the RCCL all-reduce skipped at
    world 1 -- against em.train over the one-shot E-step"""
    import ctypes as C
    import os
    from cpecan_load import ROOT
    L = C.CDLL(os.path.join(ROOT, "cpecan-signal_amd", "libcpecan_em.so"))
    L.cpecan_em_last_error.restype = C.c_char_p

    class EmInput(C.Structure):
        _fields_ = [("device", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("id_file", C.c_char_p),
                    ("items", C.c_void_p), ("n_items", C.c_int64), ("x_chars", C.c_char_p), ("n_x", C.c_int64),
                    ("events", C.c_void_p), ("n_events", C.c_int64), ("anchors", C.c_void_p),
                    ("n_anchor_pairs", C.c_int64), ("match_tables", C.POINTER(C.c_void_p)), ("n_models", C.c_int32),
                    ("gap_y_table", C.c_void_p), ("params", cp.BandParams)]

    batch = synth.make_batch(47, 6, 150, 310, anchor_every=30)
    bp = band_params(0.01, 100, 20, 40)
    n = len(batch["items"])
    items = np.zeros(n, cp.ITEM_DTYPE)
    for i, it in enumerate(batch["items"]):
        items[i] = (it["x_offset"], it["lX"], it["y_offset"], it["lY"], it["anchor_offset"], it["n_anchors"],
                    it["model"], 1, 1, 0)
    tables = [np.ascontiguousarray(m[0], dtype=np.float64) for m in batch["models"]]
    gap_y = np.ascontiguousarray(batch["models"][0][2], dtype=np.float64)
    ev = np.ascontiguousarray(batch["events"], dtype=np.float64).reshape(-1)
    an = np.ascontiguousarray(batch["anchors"], dtype=np.int64).reshape(-1)
    ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in tables])
    x = batch["x_chars"]
    inp = EmInput(0, 0, 1, None, items.ctypes.data, n, x, len(x), ev.ctypes.data, ev.size // 3, an.ctypes.data,
                  an.size // 2, ptrs, n, gap_y.ctypes.data, bp)
    trans = np.array(cp.NANOPORE_TRANSITIONS, dtype=np.float64)
    gap_x = np.array(batch["models"][0][1], dtype=np.float64)
    like = np.zeros(3)
    rc = L.cpecan_em_run(C.byref(inp), 3, C.c_double(1e-9), trans.ctypes.data_as(C.c_void_p),
                         gap_x.ctypes.data_as(C.c_void_p), like.ctypes.data_as(C.c_void_p))
    assert rc == 0, L.cpecan_em_last_error()

    ctx = cp.Context(0)
    r = dist_em.train(lambda t, g: dist_em.gpu_e_step(cp, ctx, batch, bp, list(range(n)), t, g, pseudocount=1e-9),
                      cp.NANOPORE_TRANSITIONS, batch["models"][0][1], 3)
    ctx.close()
    assert np.allclose(like, r["running_likelihoods"], rtol=1e-10) and like[0] < like[1] < like[2]
    assert np.allclose(trans[:8], r["transitions"][:8], rtol=1e-8) and np.isneginf(trans[8])
    assert np.allclose(gap_x, r["gap_x"], rtol=1e-8)
    # a world of two without a rendezvous file is refused before anything touches the GPU
    inp.world, inp.rank = 2, 1
    assert L.cpecan_em_run(C.byref(inp), 1, C.c_double(0.0), trans.ctypes.data_as(C.c_void_p),
                           gap_x.ctypes.data_as(C.c_void_p), None) != 0
    assert b"bad argument" in L.cpecan_em_last_error()

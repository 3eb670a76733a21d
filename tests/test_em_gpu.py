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

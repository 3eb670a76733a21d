"""Seeded random sweep of the banded posterior path against the oracle: sequence lengths, event/k-mer ratio,
anchor spacing, band expansion (all four builds of the systolic kernels, and the general kernel where the band is
too wide), traceback spacing and margin (including windows shorter than the margin), threshold, ragged ends.
Same bar as everywhere: totals and posterior exponents bit-identical, pairs in the reference's order."""
import numpy as np
import pytest

import synth
from harness import assert_same_pairs, band_params, cp, run_gpu, run_oracle_item

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = cp.Context(0)
    yield c
    c.close()


def cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        lX = int(rng.integers(20, 700))
        out.append(dict(seed=1000 + k, lX=lX, lY=max(8, int(lX * rng.uniform(1.2, 2.6))),
                        every=int(rng.choice([8, 20, 50, 120, 10 ** 6])),
                        e=int(rng.choice([0, 2, 10, 20, 40, 60, 100, 120, 180])),
                        md=0, tb=int(rng.integers(1, 60)),
                        thr=float(rng.choice([0.5, 0.01, 1e-4])),
                        ragged=(int(rng.integers(0, 2)), int(rng.integers(0, 2)))))
    for c in out:  # getPosteriorProbsWithBanding asserts traceBackDiagonals + 1 < minDiagsBetweenTraceBack (:880-884)
        c["md"] = c["tb"] + 2 + int(rng.integers(0, 350))
    return out


@pytest.mark.parametrize("case", cases(40, 20251004), ids=lambda c: "s%d" % c["seed"])
def test_random_case(ctx, case):
    batch = synth.make_batch(case["seed"], 2, case["lX"], case["lY"], anchor_every=case["every"])
    bp = band_params(case["thr"], case["md"], case["tb"], case["e"])
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_AUTO, ragged=case["ragged"])
    info = b.info()
    if info["kernel"] == "systolic":
        w = info["max_band_width"]
        assert info["waves_per_workgroup"] == 2 + (w > 120) + (w > 184)
    b.close()
    for i in range(2):
        ref = run_oracle_item(batch, i, bp, case["ragged"])
        assert res[i]["cells"] == ref["cells"]
        assert np.array_equal(res[i]["totals_xay"], ref["totals_xay"])
        assert np.array_equal(res[i]["totals"], ref["totals"])
        assert_same_pairs(res[i], ref)


@pytest.mark.parametrize("case", cases(10, 77), ids=lambda c: "v%d" % c["seed"])
def test_random_case_vanilla(ctx, case):
    import pyoracle as o
    import test_vanilla_gpu as tv
    batch = synth.make_batch(case["seed"], 2, min(case["lX"], 300), min(case["lY"], 700), anchor_every=case["every"])
    models = [o.VanillaModel(m, tv.skip_bins(i), gy) for i, (m, _, gy) in enumerate(batch["models"])]
    tv.run(ctx, batch, models, band_params(case["thr"], case["md"], case["tb"], case["e"]), case["ragged"])


@pytest.mark.parametrize("case", cases(10, 99), ids=lambda c: "d%d" % c["seed"])
def test_random_case_dna(ctx, case):
    import test_dna5_gpu as td
    rng = np.random.default_rng(case["seed"])
    seqs = []
    for _ in range(2):
        x, y, pairs = td.evolve(rng, min(case["lX"], 400))
        a = pairs[3::max(2, min(case["every"], 60))] if case["every"] < 10 ** 6 else np.zeros((0, 2), np.int64)
        seqs.append((x, y, a))
    td.run_case(ctx, seqs, band_params(case["thr"], case["md"], case["tb"], case["e"]), case["ragged"])

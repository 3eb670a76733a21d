"""Seeded random sweep of the banded posterior path against the oracle: sequence lengths, event/k-mer ratio,
anchor spacing, band expansion (all four builds of the systolic kernels, and the general kernel where the band is
too wide), traceback spacing and margin (including windows shorter than the margin), threshold, ragged ends.
Same bar as everywhere: totals and posterior exponents bit-identical, pairs in the reference's order."""
import os

import numpy as np
import pytest

import synth
from harness import assert_same_pairs, band_params, cp, run_gpu, run_oracle_item

pytestmark = pytest.mark.gpu

# CPECAN_FUZZ_SCALE=N runs N times as many cases of every kind (the first ones are the default run's)
SCALE = max(1, int(os.environ.get("CPECAN_FUZZ_SCALE", "1")))


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


@pytest.mark.parametrize("family", [0, cp.FLAG_WORKGROUP_KERNELS], ids=["wave", "workgroup"])
@pytest.mark.parametrize("case", cases(40 * SCALE, 20251004), ids=lambda c: "s%d" % c["seed"])
def test_random_case(ctx, case, family):
    """every case on both families of register-resident kernels (one wave / one workgroup per alignment)"""
    batch = synth.make_batch(case["seed"], 2, case["lX"], case["lY"], anchor_every=case["every"])
    bp = band_params(case["thr"], case["md"], case["tb"], case["e"])
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_AUTO, flags=family, ragged=case["ragged"])
    info = b.info()
    if info["kernel"] == "systolic":
        w = info["max_band_width"]
        assert info["family"] == ("workgroup" if family else "wave")
        if family:  # the fewest waves that hold the band: 1, 2, 3, 4 for bands up to 56, 120, 184, 248 k-mers
            assert info["waves_per_workgroup"] == 1 + (w > 56) + (w > 120) + (w > 184)
        else:       # the fewest cells per lane from 2
            assert info["waves_per_workgroup"] == 2 + (w > 120) + (w > 184)
    b.close()
    for i in range(2):
        ref = run_oracle_item(batch, i, bp, case["ragged"])
        assert res[i]["cells"] == ref["cells"]
        assert np.array_equal(res[i]["totals_xay"], ref["totals_xay"])
        assert np.array_equal(res[i]["totals"], ref["totals"])
        assert_same_pairs(res[i], ref)


def asm_cases(n, seed):
    """bands of 121-158 k-mers (diagonalExpansion 100-128 around anchors on the path): what the assembly sweeps take"""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        lX = int(rng.integers(250, 1100))
        c = dict(seed=5000 + k, lX=lX, lY=int(lX * rng.uniform(1.6, 2.4)), every=int(rng.choice([25, 50, 80])),
                 e=int(rng.choice([100, 104, 110, 120, 128])), tb=int(rng.integers(1, 60)),
                 thr=float(rng.choice([0.5, 0.01, 1e-4, 0.0])), ragged=(int(rng.integers(0, 2)), int(rng.integers(0, 2))),
                 flags=int(rng.choice([0, cp.FLAG_SMALL_FOOTPRINT])), n=int(rng.integers(1, 4)))
        c["md"] = c["tb"] + 2 + int(rng.integers(0, 500))
        out.append(c)
    return out


@pytest.mark.parametrize("case", asm_cases(24 * SCALE, 31337), ids=lambda c: "a%d" % c["seed"])
def test_random_case_assembly_sweeps(ctx, case):
    """the hand-scheduled sweeps (both layouts of their ring) wherever the batch is shaped for them; the compiled kernels
    take what is left (a band that turns out narrower or wider), to the same bar"""
    batch = synth.make_batch(case["seed"], case["n"], case["lX"], case["lY"], anchor_every=case["every"], length_sigma=0.2)
    bp = band_params(case["thr"], case["md"], case["tb"], case["e"])
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_AUTO, flags=case["flags"], ragged=case["ragged"])
    info = b.info()
    if info["kernel"] == "systolic" and info["waves_per_workgroup"] == 3 and info["max_band_width"] <= 158:
        assert info["assembly_sweeps"] == 2
    b.close()
    for i in range(case["n"]):
        ref = run_oracle_item(batch, i, bp, case["ragged"])
        assert res[i]["cells"] == ref["cells"]
        assert np.array_equal(res[i]["totals_xay"], ref["totals_xay"])
        assert np.array_equal(res[i]["totals"], ref["totals"])
        assert_same_pairs(res[i], ref)


@pytest.mark.parametrize("case", cases(10 * SCALE, 77), ids=lambda c: "v%d" % c["seed"])
def test_random_case_vanilla(ctx, case):
    import pyoracle as o
    import test_vanilla_gpu as tv
    batch = synth.make_batch(case["seed"], 2, min(case["lX"], 300), min(case["lY"], 700), anchor_every=case["every"])
    models = [o.VanillaModel(m, tv.skip_bins(i), gy) for i, (m, _, gy) in enumerate(batch["models"])]
    tv.run(ctx, batch, models, band_params(case["thr"], case["md"], case["tb"], case["e"]), case["ragged"])


@pytest.mark.parametrize("case", cases(10 * SCALE, 99), ids=lambda c: "d%d" % c["seed"])
def test_random_case_dna(ctx, case, monkeypatch):
    import test_dna5_gpu as td
    monkeypatch.setenv("CPECAN_WAVE5_PAIRED", str(case["seed"] % 2))  # one wave per alignment / a forward + a backward wave
    rng = np.random.default_rng(case["seed"])
    seqs = []
    for _ in range(2):
        x, y, pairs = td.evolve(rng, min(case["lX"], 400))
        a = pairs[3::max(2, min(case["every"], 60))] if case["every"] < 10 ** 6 else np.zeros((0, 2), np.int64)
        seqs.append((x, y, a))
    td.run_case(ctx, seqs, band_params(case["thr"], case["md"], case["tb"], case["e"]), case["ragged"])


@pytest.mark.parametrize("case", cases(14 * SCALE, 4242), ids=lambda c: "h%d" % c["seed"])
def test_random_case_hdp(ctx, case, golden_dir):
    """the HDP machine over the same random shapes: the wave-per-alignment HDP kernels where the band fits (two,
    three or four cells per lane), the general kernel where it does not -- totals, exponents and pairs identical to
    the oracle either way"""
    import os
    import pyoracle as o
    import test_hdp_gpu as th
    from harness import make_items, orc_params
    nhdp = o.load_nhdp(os.path.join(golden_dir, "testTemplate.nhdp"))
    every = min(case["every"], 200)
    batch, model = th.hdp_batch(case["seed"], 2, min(case["lX"], 450), every, nhdp)
    ctx.models_clear()
    ctx.modelsh_create([(cp.NANOPORE_TRANSITIONS, nhdp["alphabet"], nhdp["grid"], nhdp["y"], nhdp["slope"],
                         nhdp["kmer_row"])])
    bp = band_params(case["thr"], case["md"], case["tb"], case["e"])
    b = cp.Batch(ctx, make_items(batch, case["ragged"]), batch["x_chars"], batch["events"], batch["anchors"], bp,
                 hdp=True)
    info = b.info()
    if info["max_band_width"] <= 248:
        assert info["kernel"] == "systolic" and info["family"] == "wave"
    b.run()
    b.sync()
    npairs, ntot, ncells = b.counts()
    p = orc_params(bp, split=1 << 60)
    for i, it in enumerate(batch["items"]):
        x = batch["x_chars"][it["x_offset"]: it["x_offset"] + it["lX"] + 5]
        ev = batch["events"][it["y_offset"]: it["y_offset"] + it["lY"]]
        an = batch["anchors"][it["anchor_offset"]: it["anchor_offset"] + it["n_anchors"]]
        tri, lp = b.pairs(i, npairs[i])
        xay, tot = b.totals(i, ntot[i])
        ref = o.aligned_pairs_using_anchors(model, x, it["lX"], ev, an, p, case["ragged"][0], case["ragged"][1])
        ref["triples"], ref["logp"] = ref["triples"][::-1], ref["logp"][::-1]
        assert int(ncells[i]) == ref["cells"]
        assert np.array_equal(xay, ref["totals_xay"])
        assert np.array_equal(tot, ref["totals"])
        assert_same_pairs(dict(triples=tri, logp=lp), ref)
    b.close()

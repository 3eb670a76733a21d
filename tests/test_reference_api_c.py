"""tests/c/reference_api_test.c: a C program compiled against include/cpecan_api.h alone that uses the API the way
the reference's callers and CuTest suites do (struct members, function pointers, Hmm subclasses).  CPU: the host
internals reproduce the reference's known answers.  GPU: the call sequence of getSignalExpectations
(vanillaAlign.c:318-359) against the oracle, and the exported internals against the aligner entry points."""
import os
import subprocess

import numpy as np
import pytest

import pyoracle as o
from cpecan_load import ROOT

SRC = os.path.join(ROOT, "tests", "c", "reference_api_test.c")
LIBDIR = os.path.join(ROOT, "cpecan-signal_amd")


@pytest.fixture(scope="module")
def program(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cprog") / "reference_api_test")
    subprocess.run(["gcc", "-O1", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    SRC, "-o", exe, "-L" + LIBDIR, "-lcpecan_host", "-lcpecan_hip", "-Wl,-rpath," + LIBDIR, "-lm", "-lpthread"],
                   check=True)
    return exe


def _run(args, timeout=600):
    r = subprocess.run(args, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_reference_shaped_api_on_the_host(program, golden_dir):
    out = _run([program, "cpu", golden_dir])
    names = [l.split()[1] for l in out.splitlines() if l.startswith("ok ")]
    assert "FAILED" not in out
    assert names == ["stateMachine_members", "small_helpers", "strawMan_cell", "vanilla_cell", "fiveState_cell", "hdp_density",
                     "dpDiagonal_dpMatrix", "fiveState_diagonalDPCalculations", "strawMan_diagonalDPCalculations",
                     "vanilla_diagonalDPCalculations", "plugin_constructors", "continuousPairHmm", "vanillaHmm",
                     "hdpHmm"]
    # the host density equals the oracle's (same spline, same row)
    want = o.HdpModel(o.load_nhdp(os.path.join(golden_dir, "testTemplate.nhdp"))).density("ATGACA", 60.032615)
    got = float([l for l in out.splitlines() if l.startswith("ok hdp_density")][0].split()[2])
    assert got == want and got > 0


def _anchors_from(triples, every=30, drop=range(8, 15)):
    best = {}
    for q, x, y in triples:
        if q > 9000000:
            best[int(x)] = int(y)
    anchors, py = [], -1
    for x in sorted(best)[::every]:
        if best[x] > py:
            anchors.append((x, best[x]))
            py = best[x]
    return [a for i, a in enumerate(anchors) if i not in drop]


def _write_inputs(d, target, events, anchors_xy, event_map):
    """anchors go in 'unmapped' (x, read position), as vanillaAlign gets them from lastz; the event map takes the read
    position to the event (nanopore_remapAnchorPairsWithOffset)"""
    event_map = np.ascontiguousarray(event_map, dtype=np.int64)
    unmapped, kept = [], []
    for x, y in anchors_xy:
        r = int(np.searchsorted(event_map, y))
        if r < event_map.size and event_map[r] == y:
            unmapped.append((x, r))
            kept.append((x, y))
    paths = {k: os.path.join(d, k) for k in ("target.txt", "events.f64", "anchors.txt", "map.i64", "out.hmm")}
    open(paths["target.txt"], "w").write(target + "\n")
    np.ascontiguousarray(events, dtype=np.float64).reshape(-1).tofile(paths["events.f64"])
    open(paths["anchors.txt"], "w").write("".join("%d %d\n" % a for a in unmapped))
    event_map.tofile(paths["map.i64"])
    return paths, kept


def _estep(program, sm_type, model, paths, params5, threshold):
    out = _run([program, "estep", str(sm_type), model, paths["target.txt"], paths["events.f64"], paths["anchors.txt"],
                paths["map.i64"]] + ["%.17g" % v for v in params5] + ["%.17g" % threshold, paths["out.hmm"]])
    rec = {}
    for line in out.splitlines():
        k, *v = line.split()
        rec.setdefault(k, []).append(v)
    return rec


PARAMS = dict(minDiagsBetweenTraceBack=150, diagonalExpansion=40, splitMatrixBiggerThanThis=100 * 100)


@pytest.mark.gpu
def test_get_signal_expectations_three_state(program, golden_dir, zymo_read, template_model, tmp_path):
    model = os.path.join(golden_dir, "template_median68pA.model")
    ref_seq, ev = zymo_read["reference"], zymo_read["template_events"]
    lX = len(ref_seq) - 5
    om = o.Sm3Model(template_model[0], template_model[2]).scaled(*zymo_read["template_params"])
    unbanded = o.aligned_pairs_without_banding(om, ref_seq, lX, ev, o.default_params())
    paths, anchors = _write_inputs(str(tmp_path), ref_seq, ev, _anchors_from(unbanded["triples"]),
                                   zymo_read["template_map"])
    assert len(anchors) > 10
    rec = _estep(program, 2, model, paths, zymo_read["template_params"], 0.01)
    assert int(rec["anchors"][0][0]) == len(anchors)
    want = o.OrcExpectations()
    o.aligned_pairs_using_anchors(om, ref_seq, lX, ev, anchors, o.default_params(**PARAMS), True, True,
                                  expectations=want)
    assert np.allclose(np.array(rec["transitions"][0], float), np.array(want.transitions[:]), rtol=1e-9)
    assert np.allclose(np.array(rec["kmergap"][0], float), np.array(want.kmerGap[:]), rtol=1e-9, atol=1e-300)
    assert np.isclose(float(rec["likelihood"][0][0]), want.likelihood, rtol=1e-12)
    # the file hmmContinuous_writeToFile left: type, states, symbols; 9 transitions + likelihood; 4096 k-mer gaps
    lines = open(paths["out.hmm"]).read().split("\n")
    assert lines[0].split() == ["2", "3", "4096"] and len(lines[1].split()) == 10 and len(lines[2].split()) == 4096
    assert np.allclose(np.array(lines[1].split()[:9], float), np.array(want.transitions[:]), atol=1e-6)


@pytest.mark.gpu
def test_get_signal_expectations_vanilla(program, golden_dir, zymo_read, template_model, tmp_path):
    model = os.path.join(golden_dir, "template_median68pA.model")
    ref_seq, ev = zymo_read["reference"], zymo_read["template_events"]
    lX = len(ref_seq) - 5
    match, skip, gapy = template_model
    om = o.VanillaModel(match, skip, gapy, float(np.float32(0.17)), float(np.float32(0.55))).scaled(
        *zymo_read["template_params"])
    unbanded = o.aligned_pairs_without_banding(om, ref_seq, lX, ev, o.default_params(threshold=0.2))
    paths, anchors = _write_inputs(str(tmp_path), ref_seq, ev, _anchors_from(unbanded["triples"]),
                                   zymo_read["template_map"])
    assert len(anchors) > 10
    rec = _estep(program, 4, model, paths, zymo_read["template_params"], 0.01)
    want = o.expectations_v_using_anchors(om, ref_seq, lX, ev, anchors, o.default_params(**PARAMS),
                                          o.OrcExpectationsV(), True, True)
    got = np.array(rec["bins"][0], float)
    assert np.allclose(got, list(want.kmerSkipBins), rtol=1e-9, atol=1e-12) and np.count_nonzero(got) > 30
    assert np.isclose(float(rec["likelihood"][0][0]), want.likelihood, rtol=1e-12)
    lines = open(paths["out.hmm"]).read().split("\n")  # header, 60 bins + likelihood, the two implanted tables
    assert lines[0].split()[0] == "4" and len(lines[1].split()) == 61
    assert len(lines[2].split()) == 1 + 4096 * 5 and len(lines[3].split()) == 1 + 4096 * 5


@pytest.mark.gpu
def test_get_signal_expectations_hdp(program, golden_dir, tmp_path):
    nhdp = os.path.join(golden_dir, "testTemplate.nhdp")
    parsed = o.load_nhdp(nhdp)
    om = o.HdpModel(parsed)
    rng = np.random.default_rng(71)
    lX = 160
    x = "".join(rng.choice(list("ACGT"), lX + 5))
    ev, anchors = [], []
    for k in range(lX):
        row = parsed["kmer_row"][om.kmer_id(x[k:k + 6])]
        mode = parsed["grid"][int(np.argmax(parsed["y"][row]))]
        if k % 40 == 20:
            anchors.append((k, len(ev)))
        for _ in range(1 if rng.random() < 0.6 else 2):
            ev.append((mode + rng.normal(0, 1.0), 1.0, 0.01))
    ev = np.array(ev)
    paths, kept = _write_inputs(str(tmp_path), x, ev, anchors, np.arange(len(ev) + 1))
    assert kept == anchors
    rec = _estep(program, 7, nhdp, paths, [1, 0, 1, 1, 1], 0.05)
    want = o.expectations_h_using_anchors(om, [(x, lX, ev, anchors)], o.default_params(threshold=0.05, **PARAMS),
                                          0.05, True, True)
    assert np.allclose(np.array(rec["transitions"][0], float), want["transitions"], rtol=1e-9, atol=1e-12)
    assert np.isclose(float(rec["likelihood"][0][0]), want["likelihood"], rtol=1e-12)
    n = int(rec["assignments"][0][0])
    assert n == len(want["assign"]) and n > 20
    # k-mer, event mean, and where the two pointers sit in the caller's sequences
    got = [(a[0], float(a[1]), int(a[2]), int(a[3])) for a in rec["assign"]]
    assert got == [(x[int(ix):int(ix) + 6], ev[int(iy), 0], int(ix), int(iy)) for _, ix, iy in want["assign"]]
    lines = open(paths["out.hmm"]).read().split("\n")
    assert lines[0].split() == ["7", "3", "0.050000", str(n)] and lines[3].split() == [g[0] for g in got]


@pytest.mark.gpu
def test_exported_internals_against_the_gpu_path(program, golden_dir):
    out = _run([program, "gpu", golden_dir])
    assert "FAILED" not in out
    assert [l.split()[1] for l in out.splitlines() if l.startswith("ok ")] == [
        "strawMan_host_vs_gpu", "vanilla_host_vs_gpu", "fiveState_host_vs_gpu", "two_threads_two_machines"]

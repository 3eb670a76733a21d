"""The reference-shaped host library (include/cpecan_api.h -> libcpecan_host.so): exports and host-side
integer logic on CPU; the alignment entry points against the oracle on the GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pyoracle as o
import synth
from cpecan_load import ROOT
import host_api as h


def test_host_library_exports_declared_symbols():
    """every function include/cpecan_api.h declares is exported by libcpecan_host.so"""
    header = open(os.path.join(ROOT, "include", "cpecan_api.h")).read()
    declared = set(re.findall(r"\b([A-Za-z_][A-Za-z_0-9]*)\s*\(", re.sub(r"/\*.*?\*/", "", header, flags=re.S)))
    declared -= {"_Static_assert", "offsetof", "sizeof", "LOG_ZERO", "double", "int64_t", "void", "bool", "int", "char"}
    assert len(declared) > 180 and set(h.EXPORTS) <= declared, set(h.EXPORTS) - declared
    lib = C.CDLL(h.LIB_PATH)
    missing = sorted(name for name in declared if not hasattr(lib, name))
    assert not missing, missing


def test_host_defaults_and_split_points_golden():
    L = h.lib()
    p = L.pairwiseAlignmentBandingParameters_construct()
    assert (p.contents.threshold, p.contents.minDiagsBetweenTraceBack, p.contents.traceBackDiagonals,
            p.contents.diagonalExpansion, p.contents.splitMatrixBiggerThanThis) == (0.01, 1000, 40, 20, 9000000)
    L.pairwiseAlignmentBandingParameters_destruct(p)
    # tests/pairwiseAlignerTest.c:634-661
    anchors = [(2000, 2000), (4002, 4001), (5000, 5000), (8000, 6000), (9000, 9000), (10000, 14000),
               (15000, 15000), (16000, 16000)]
    lst = h.make_anchor_list(anchors)
    sp = L.getSplitPoints(lst, 20000, 25000, 2000 * 2000, False, False)
    assert h.list_to_array(sp, 4).tolist() == [
        [0, 0, 3001, 3001], [3002, 3001, 9500, 11001], [9501, 12000, 12001, 14500],
        [13000, 14501, 18000, 18001], [18001, 23000, 20000, 25000]]
    L.stList_destruct(sp)
    L.stList_destruct(lst)
    assert L.emissions_discrete_getKmerIndex(b"AAAAAC") == 1
    assert L.emissions_discrete_getKmerIndex(b"TTTTTT") == 4095
    assert L.emissions_discrete_getKmerIndex(b"ACNTAA") > 4096
    assert L.sequence_correctSeqLength(13, 2) == 8


def test_model_loader_and_scaling_match_reference_semantics(golden_dir, zymo_read, template_model):
    L = h.lib()
    sm = L.getStrawManStateMachine3(os.path.join(golden_dir, "template_median68pA.model").encode())
    match = np.ctypeslib.as_array(sm.contents.model.EMISSION_MATCH_PROBS, shape=(1 + 4096 * 5,))
    assert np.array_equal(match, template_model[0])
    assert sm.contents.TRANSITION_GAP_OPEN_Y == -4.3187242127300092
    L.emissions_signal_scaleModel(sm, *zymo_read["template_params"])
    ref = template_model[0].copy()
    o.lib().orc_scale_model(ref.ctypes.data, *zymo_read["template_params"])
    assert np.array_equal(match, ref)  # tests/signalPairwiseTest.c:1007-1040, exact
    L.stateMachine_destruct(sm)


@pytest.mark.gpu
def test_zymo_read_through_host_api(golden_dir, zymo_read, template_model):
    """vanillaAlign-style use of the API on the reference's own read: 986 un-banded pairs
    (signalPairwiseTest.c:1166-1173) and a banded, split alignment identical to the oracle's."""
    L = h.lib()
    sm = L.getStrawManStateMachine3(os.path.join(golden_dir, "template_median68pA.model").encode())
    L.emissions_signal_scaleModel(sm, *zymo_read["template_params"])
    rd = h.Read(zymo_read["reference"], zymo_read["template_events"])
    p = L.pairwiseAlignmentBandingParameters_construct()
    pairs = L.getAlignedPairsWithoutBanding(sm, C.cast(rd.xbuf, C.c_void_p), rd.ev.ctypes.data_as(C.c_void_p),
                                            rd.lX, rd.lY, p, h.fn_ptr("sequence_getKmer"),
                                            h.fn_ptr("sequence_getEvent"),
                                            h.fn_ptr("diagonalCalculationPosteriorMatchProbs"), False, False)
    got = h.list_to_array(pairs)
    L.stList_destruct(pairs)
    assert len(got) == 986
    om = o.Sm3Model(template_model[0], template_model[2]).scaled(*zymo_read["template_params"])
    ref = o.aligned_pairs_without_banding(om, zymo_read["reference"], rd.lX, zymo_read["template_events"],
                                          o.default_params())
    assert np.array_equal(got, ref["triples"])  # same pairs, same order, same integer posteriors

    # banded, with anchors from the un-banded result and a small split threshold => several sub-alignments
    best = {}
    for q, x, y in ref["triples"]:
        if q > 9000000:
            best[int(x)] = int(y)
    anchors, py = [], -1
    for x in sorted(best)[::30]:
        if best[x] > py:
            anchors.append((x, best[x]))
            py = best[x]
    anchors = [a for i, a in enumerate(anchors) if not 8 <= i <= 14]   # a big anchor-free gap
    p.contents.splitMatrixBiggerThanThis = 100 * 100
    p.contents.diagonalExpansion = 40
    p.contents.minDiagsBetweenTraceBack = 150
    lst = h.make_anchor_list(anchors)
    pairs = L.getAlignedPairsUsingAnchors(sm, rd.sX, rd.sY, lst, p,
                                          h.fn_ptr("diagonalCalculationPosteriorMatchProbs"), True, True)
    got = h.list_to_array(pairs)
    L.stList_destruct(pairs)
    op = o.default_params(minDiagsBetweenTraceBack=150, diagonalExpansion=40,
                          splitMatrixBiggerThanThis=100 * 100)
    assert len(o.split_points(anchors, rd.lX, rd.lY, 100 * 100, 1, 1)) > 1
    ref = o.aligned_pairs_using_anchors(om, zymo_read["reference"], rd.lX, zymo_read["template_events"],
                                        anchors, op, True, True)
    assert np.array_equal(got, ref["triples"])  # same pairs, same order, same integer posteriors

    # expectations for Baum-Welch on the same alignment
    e = h.Expectations()
    L.cpecan_getSignalExpectationsUsingAnchors(sm, C.byref(e), rd.sX, rd.sY, lst, p, True, True)
    oe = o.OrcExpectations()
    o.aligned_pairs_using_anchors(om, zymo_read["reference"], rd.lX, zymo_read["template_events"], anchors,
                                  op, True, True, expectations=oe)
    assert np.allclose(np.array(e.transitions[:]), np.array(oe.transitions[:]), rtol=1e-9)
    assert np.allclose(np.array(e.individualKmerGapProbs[:]), np.array(oe.kmerGap[:]), rtol=1e-9, atol=1e-300)
    assert np.isclose(e.likelihood, oe.likelihood, rtol=1e-12)
    L.cpecan_pairHmmExpectations_normalize(C.byref(e))
    assert np.allclose(np.array(e.transitions[:]).reshape(3, 3).sum(1), 1.0)
    L.stList_destruct(lst)
    L.pairwiseAlignmentBandingParameters_destruct(p)
    L.stateMachine_destruct(sm)


@pytest.mark.gpu
def test_config1_one_5k_event_template_read_against_2kb(golden_dir, template_model):
    """BASELINE.json configs[1]: the 3-state signal machine with the real template_median68pA model, scaled for the
    read as vanillaAlign does, one ~5 000-event template read against a 2 kb reference, banded (diagonalExpansion 50,
    vanillaAlign.c:377), ragged ends -- through the reference's entry point, identical to the oracle: same pairs,
    same order, same integer posteriors."""
    L = h.lib()
    match, _, gapy = template_model
    rd = synth.make_read(np.random.default_rng(20260), match, 2000, 5000, anchor_every=50)
    sm = L.getStrawManStateMachine3(os.path.join(golden_dir, "template_median68pA.model").encode())
    L.emissions_signal_scaleModel(sm, *rd["scale_params"])
    read = h.Read(rd["seq"], rd["events"])
    assert (read.lX, read.lY) == (2000, 5000)
    p = L.pairwiseAlignmentBandingParameters_construct()
    p.contents.diagonalExpansion = 50
    lst = h.make_anchor_list(rd["anchors"])
    pairs = L.getAlignedPairsUsingAnchors(sm, read.sX, read.sY, lst, p,
                                          h.fn_ptr("diagonalCalculationPosteriorMatchProbs"), True, True)
    got = h.list_to_array(pairs)
    L.stList_destruct(pairs)
    om = o.Sm3Model(match, gapy).scaled(*rd["scale_params"])
    ref = o.aligned_pairs_using_anchors(om, rd["seq"].decode(), 2000, rd["events"], rd["anchors"],
                                        o.default_params(diagonalExpansion=50), True, True)
    assert len(got) > 3000 and ref["cells"] > 500000
    assert np.array_equal(got, ref["triples"])
    # and its E-step through the reference's own container
    hmm = L.hmmContinuous_getEmptyHmm(2, 0.0, 0.0)
    L.getExpectationsUsingAnchors(sm, hmm, read.sX, read.sY, lst, p, h.fn_ptr("diagonalCalculation_Expectations"),
                                  True, True)
    want = o.OrcExpectations()
    o.aligned_pairs_using_anchors(om, rd["seq"].decode(), 2000, rd["events"], rd["anchors"],
                                  o.default_params(diagonalExpansion=50), True, True, expectations=want)
    cp = C.cast(hmm, C.POINTER(h.ContinuousPairHmm)).contents
    assert np.allclose([cp.transitions[i] for i in range(9)], np.array(want.transitions[:]), rtol=1e-9)
    assert np.allclose([cp.individualKmerGapProbs[i] for i in range(4096)], np.array(want.kmerGap[:]), rtol=1e-9,
                       atol=1e-300)
    assert np.isclose(cp.baseHmm.likelihood, want.likelihood, rtol=1e-12)
    L.hmmContinuous_destruct(hmm, 2)
    L.stList_destruct(lst)
    L.pairwiseAlignmentBandingParameters_destruct(p)
    L.stateMachine_destruct(sm)


@pytest.mark.gpu
def test_batch_entry_matches_single_calls(golden_dir):
    L = h.lib()
    batch = synth.make_batch(31, 5, 150, 310, anchor_every=30, distinct_models=False)
    sm = L.getStrawManStateMachine3(None)
    m, gx, gy = batch["models"][0]
    C.memmove(sm.contents.model.EMISSION_MATCH_PROBS, m.ctypes.data, m.nbytes)
    C.memmove(sm.contents.model.EMISSION_GAP_Y_PROBS, gy.ctypes.data, gy.nbytes)
    p = L.pairwiseAlignmentBandingParameters_construct()
    p.contents.diagonalExpansion = 40
    p.contents.minDiagsBetweenTraceBack = 120
    reads, lists = [], []
    for it in batch["items"]:
        x = batch["x_chars"][it["x_offset"]: it["x_offset"] + it["lX"] + 5]
        ev = batch["events"][it["y_offset"]: it["y_offset"] + it["lY"]]
        reads.append(h.Read(x, ev))
        lists.append(h.make_anchor_list(batch["anchors"][it["anchor_offset"]: it["anchor_offset"] + it["n_anchors"]]))
    n = len(reads)
    vp = C.c_void_p
    sms = (vp * n)(*[C.cast(sm, vp)] * n)
    sxs = (vp * n)(*[r.sX for r in reads])
    sys_ = (vp * n)(*[r.sY for r in reads])
    ans = (vp * n)(*lists)
    out = L.getAlignedPairsUsingAnchorsBatch(n, sms, sxs, sys_, ans, p, True, True)
    for i in range(n):
        a = h.list_to_array(out[i])
        single = L.getAlignedPairsUsingAnchors(sm, reads[i].sX, reads[i].sY, lists[i], p,
                                               h.fn_ptr("diagonalCalculationPosteriorMatchProbs"), True, True)
        b = h.list_to_array(single)
        assert np.array_equal(a, b) and len(a) > 0
        L.stList_destruct(single)
        L.stList_destruct(out[i])


@pytest.mark.gpu
def test_five_state_dna_through_host_api():
    """the reference's DNA-against-DNA use of the API (tests/pairwiseAlignerTest.c:503-560 style):
    stateMachine5_construct + sequence_getBase + getAlignedPairsUsingAnchors, against the oracle."""
    L = h.lib()
    rng = np.random.default_rng(5)
    x = "".join(rng.choice(list("ACGT"), 180))
    y = "".join(ch if rng.random() > 0.15 else rng.choice(list("ACGT")) for ch in x[:70] + x[78:])
    sm = L.stateMachine5_construct(0, 4, h.fn_ptr("emissions_symbol_setEmissionsToDefaults"),
                                   h.fn_ptr("emissions_symbol_getGapProb"), h.fn_ptr("emissions_symbol_getGapProb"),
                                   h.fn_ptr("emissions_symbol_getMatchProb"), h.fn_ptr("cell_updateExpectations"))
    xb, yb = C.create_string_buffer(x.encode()), C.create_string_buffer(y.encode())
    sX = L.sequence_construct2(len(x), C.cast(xb, C.c_void_p), h.fn_ptr("sequence_getBase"),
                               h.fn_ptr("sequence_sliceNucleotideSequence"))
    sY = L.sequence_construct2(len(y), C.cast(yb, C.c_void_p), h.fn_ptr("sequence_getBase"),
                               h.fn_ptr("sequence_sliceNucleotideSequence"))
    p = L.pairwiseAlignmentBandingParameters_construct()
    p.contents.minDiagsBetweenTraceBack = 60
    p.contents.traceBackDiagonals = 10
    anchors = [(20, 20), (60, 60), (120, 112)]
    lst = h.make_anchor_list(anchors)
    pairs = L.getAlignedPairsUsingAnchors(sm, sX, sY, lst, p, h.fn_ptr("diagonalCalculationPosteriorMatchProbs"),
                                          False, False)
    got = h.list_to_array(pairs)
    L.stList_destruct(pairs)
    op = o.default_params(minDiagsBetweenTraceBack=60, traceBackDiagonals=10)
    ref = o.aligned_pairs_using_anchors(o.Sm5Model(), x, len(x), y, anchors, op, False, False)
    assert len(got) > 100
    assert np.array_equal(got, ref["triples"])  # same pairs, same order, same integer posteriors

    # getAlignedPairs with the caller's anchor function (the reference passes its lastz wrapper there), and the
    # split driver called directly with a coordinate-correction callback, as getAlignedPairsUsingAnchors does
    anchor_fn = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(lambda cx, cy, pp: h.make_anchor_list(anchors))
    pairs = L.getAlignedPairs(sm, C.cast(xb, C.c_void_p), C.cast(yb, C.c_void_p), len(x), len(y), p,
                              h.fn_ptr("sequence_getBase"), h.fn_ptr("sequence_getBase"), anchor_fn, False, False)
    assert np.array_equal(h.list_to_array(pairs), got)
    L.stList_destruct(pairs)
    p.contents.splitMatrixBiggerThanThis = 30 * 30
    whole = L.getAlignedPairsUsingAnchors(sm, sX, sY, lst, p, h.fn_ptr("diagonalCalculationPosteriorMatchProbs"),
                                          False, False)
    sub_list, offsets = L.stList_construct3(0, h.fn_ptr("stIntTuple_destruct")), []
    corr = C.CFUNCTYPE(None, C.c_int64, C.c_int64, C.c_void_p)(lambda ox, oy, extra: offsets.append(
        (ox, oy, L.stList_length(sub_list))))
    extra = (C.c_void_p * 2)(sub_list, None)
    L.getPosteriorProbsWithBandingSplittingAlignmentsByLargeGaps(
        sm, lst, sX, sY, p, False, False, h.fn_ptr("diagonalCalculationPosteriorMatchProbs"), corr, extra)
    assert len(offsets) == len(o.split_points(anchors, len(x), len(y), 30 * 30, 0, 0)) > 1
    raw = h.list_to_array(sub_list)  # sub-alignment coordinates, emission order, region after region
    rebuilt, start = [], 0
    for ox, oy, end in offsets:  # what alignedPairCoordinateCorrectionFn does: shift, then pop from the tail
        part = raw[start:end].copy()
        part[:, 1] += ox
        part[:, 2] += oy
        rebuilt.append(part[::-1])
        start = end
    assert np.array_equal(np.concatenate(rebuilt), h.list_to_array(whole))
    L.stList_destruct(sub_list)
    L.stList_destruct(whole)
    L.stList_destruct(lst)
    L.sequence_sequenceDestroy(sX)
    L.sequence_sequenceDestroy(sY)
    L.pairwiseAlignmentBandingParameters_destruct(p)
    L.stateMachine_destruct(sm)


@pytest.mark.gpu
def test_four_state_machine_through_host_api(golden_dir, zymo_read, template_model):
    """test_stateMachine4_getAlignedPairsWithBanding (tests/signalPairwiseTest.c:1180-1245) through the host API:
    getStateMachine4 + scaleModel, un-banded at the default threshold: exactly 988 pairs, identical to the oracle;
    then banded with the oracle's anchors-free band of the shipped read replaced by anchors every 50 k-mers on the
    un-banded result (lastz is not here): identical to the oracle as well."""
    L = h.lib()
    sm = L.getStateMachine4(os.path.join(golden_dir, "template_median68pA.model").encode())
    L.emissions_signal_scaleModel(sm, *zymo_read["template_params"])
    ref_seq = zymo_read["reference"]
    xbuf = C.create_string_buffer(ref_seq.encode())
    ev = np.ascontiguousarray(zymo_read["template_events"], dtype=np.float64).reshape(-1)
    lX, lY = len(ref_seq) - 5, ev.size // 3
    p = L.pairwiseAlignmentBandingParameters_construct()
    pairs = L.getAlignedPairsWithoutBanding(sm, C.cast(xbuf, C.c_void_p), ev.ctypes.data_as(C.c_void_p), lX, lY, p,
                                            h.fn_ptr("sequence_getKmer"), h.fn_ptr("sequence_getEvent"),
                                            h.fn_ptr("diagonalCalculationPosteriorMatchProbs"), True, True)
    got = h.list_to_array(pairs)
    L.stList_destruct(pairs)
    assert len(got) == 988
    match, _, gapy = template_model
    om = o.Sm4Model(match, gapy).scaled(*zymo_read["template_params"])
    ref = o.aligned_pairs_without_banding(om, ref_seq, lX, zymo_read["template_events"], o.default_params(), 1, 1)
    assert np.array_equal(got, ref["triples"])
    # banded, anchored on every 50th pair of the un-banded alignment with posterior >= 0.9
    strong = got[got[:, 0] >= 9000000]
    anchors = np.ascontiguousarray(strong[np.argsort(strong[:, 1])][::50, 1:3])
    anchors = anchors[np.concatenate([[True], (np.diff(anchors[:, 0]) > 0) & (np.diff(anchors[:, 1]) > 0)])]
    lst = h.make_anchor_list(anchors)
    sX = L.sequence_construct2(lX, C.cast(xbuf, C.c_void_p), h.fn_ptr("sequence_getKmer"),
                               h.fn_ptr("sequence_sliceNucleotideSequence2"))
    sY = L.sequence_construct2(lY, ev.ctypes.data_as(C.c_void_p), h.fn_ptr("sequence_getEvent"),
                               h.fn_ptr("sequence_sliceEventSequence2"))
    banded = L.getAlignedPairsUsingAnchors(sm, sX, sY, lst, p, h.fn_ptr("diagonalCalculationPosteriorMatchProbs"), True, True)
    gotb = h.list_to_array(banded)
    refb = o.aligned_pairs_using_anchors(om, ref_seq, lX, zymo_read["template_events"], anchors, o.default_params(), 1, 1)
    assert np.array_equal(gotb, refb["triples"])
    assert 900 < len(gotb) < 1100
    L.stList_destruct(banded)
    L.stList_destruct(lst)
    L.sequence_sequenceDestroy(sX)
    L.sequence_sequenceDestroy(sY)
    L.pairwiseAlignmentBandingParameters_destruct(p)
    L.stateMachine_destruct(sm)


@pytest.mark.gpu
def test_vanilla_zymo_read_through_host_api(golden_dir, zymo_read, template_model):
    """test_vanilla_strandAlignmentNoBanding (tests/signalPairwiseTest.c:1042-1075) through the host API:
    getSignalStateMachine3Vanilla + scaleModel + sequence_getKmer2, un-banded, then banded with anchors and
    split sub-alignments; both identical to the oracle."""
    L = h.lib()
    sm = L.getSignalStateMachine3Vanilla(os.path.join(golden_dir, "template_median68pA.model").encode())
    L.emissions_signal_scaleModel(sm, *zymo_read["template_params"])
    ref_seq = zymo_read["reference"]
    xbuf = C.create_string_buffer(ref_seq.encode())
    ev = np.ascontiguousarray(zymo_read["template_events"], dtype=np.float64).reshape(-1)
    lX, lY = len(ref_seq) - 5, ev.size // 3
    p = L.pairwiseAlignmentBandingParameters_construct()
    # test_vanilla_getAlignedPairsWithBanding (tests/signalPairwiseTest.c:1295-1303): the machine as constructed
    # (no strand call), default parameters, un-banded: exactly 953 pairs
    pairs = L.getAlignedPairsWithoutBanding(sm, C.cast(xbuf, C.c_void_p), ev.ctypes.data_as(C.c_void_p), lX, lY, p,
                                            h.fn_ptr("sequence_getKmer2"), h.fn_ptr("sequence_getEvent"),
                                            h.fn_ptr("diagonalCalculationPosteriorMatchProbs"), False, False)
    got = h.list_to_array(pairs)
    L.stList_destruct(pairs)
    assert len(got) == 953
    om0 = o.VanillaModel(*template_model, 0.17, float(np.float32(0.55))).scaled(*zymo_read["template_params"])
    ref = o.aligned_pairs_without_banding(om0, ref_seq, lX, zymo_read["template_events"], o.default_params())
    assert np.array_equal(got, ref["triples"])
    L.stateMachine3Vanilla_setStrandTransitionsToDefaults(sm, 0)
    p.contents.threshold = 0.2
    pairs = L.getAlignedPairsWithoutBanding(sm, C.cast(xbuf, C.c_void_p), ev.ctypes.data_as(C.c_void_p), lX, lY, p,
                                            h.fn_ptr("sequence_getKmer2"), h.fn_ptr("sequence_getEvent"),
                                            h.fn_ptr("diagonalCalculationPosteriorMatchProbs"), False, False)
    got = h.list_to_array(pairs)
    L.stList_destruct(pairs)
    match, skip, gapy = template_model
    om = o.VanillaModel(match, skip, gapy, float(np.float32(0.17)), float(np.float32(0.55))).scaled(
        *zymo_read["template_params"])
    ref = o.aligned_pairs_without_banding(om, ref_seq, lX, zymo_read["template_events"],
                                          o.default_params(threshold=0.2))
    assert len(got) > 500
    assert np.array_equal(got, ref["triples"])  # same pairs, same order, same integer posteriors

    best = {}
    for q, x, y in ref["triples"]:
        if q > 9000000:
            best[int(x)] = int(y)
    anchors, py = [], -1
    for x in sorted(best)[::30]:
        if best[x] > py:
            anchors.append((x, best[x]))
            py = best[x]
    anchors = [a for i, a in enumerate(anchors) if not 8 <= i <= 14]
    p.contents.threshold = 0.01
    p.contents.splitMatrixBiggerThanThis = 100 * 100
    p.contents.diagonalExpansion = 40
    p.contents.minDiagsBetweenTraceBack = 150
    sX = L.sequence_construct2(lX, C.cast(xbuf, C.c_void_p), h.fn_ptr("sequence_getKmer2"),
                               h.fn_ptr("sequence_sliceNucleotideSequence2"))
    sY = L.sequence_construct2(lY, ev.ctypes.data_as(C.c_void_p), h.fn_ptr("sequence_getEvent"),
                               h.fn_ptr("sequence_sliceEventSequence2"))
    lst = h.make_anchor_list(anchors)
    pairs = L.getAlignedPairsUsingAnchors(sm, sX, sY, lst, p, h.fn_ptr("diagonalCalculationPosteriorMatchProbs"),
                                          True, True)
    got = h.list_to_array(pairs)
    L.stList_destruct(pairs)
    op = o.default_params(minDiagsBetweenTraceBack=150, diagonalExpansion=40, splitMatrixBiggerThanThis=100 * 100)
    assert len(o.split_points(anchors, lX, lY, 100 * 100, 1, 1)) > 1
    ref = o.aligned_pairs_using_anchors(om, ref_seq, lX, zymo_read["template_events"], anchors, op, True, True)
    assert np.array_equal(got, ref["triples"])  # same pairs, same order, same integer posteriors

    # the E-step of the same alignment (split sub-alignments included), then the M-step of the skip bins
    hmm = h.VanillaExpectations()
    L.cpecan_getVanillaExpectationsUsingAnchors(sm, C.byref(hmm), sX, sY, lst, p, True, True)
    want = o.expectations_v_using_anchors(om, ref_seq, lX, zymo_read["template_events"], anchors, op,
                                          o.OrcExpectationsV(), True, True)
    assert np.allclose(list(hmm.kmerSkipBins), list(want.kmerSkipBins), rtol=1e-9, atol=1e-12)
    assert np.isclose(hmm.likelihood, want.likelihood, rtol=1e-12) and hmm.likelihood < 0
    assert np.count_nonzero(list(hmm.kmerSkipBins)) > 30
    L.cpecan_vanillaExpectations_normalize(C.byref(hmm))
    assert abs(sum(hmm.kmerSkipBins) - 1.0) < 1e-12
    L.cpecan_vanillaExpectations_load(sm, C.byref(hmm))
    assert C.cast(sm, C.POINTER(h.StateMachine)).contents.EMISSION_GAP_X_PROBS[31] == hmm.kmerSkipBins[31]
    L.stList_destruct(lst)
    L.sequence_sequenceDestroy(sX)
    L.sequence_sequenceDestroy(sY)
    L.pairwiseAlignmentBandingParameters_destruct(p)
    L.stateMachine_destruct(sm)


@pytest.mark.gpu
def test_baum_welch_iterations_on_the_zymo_read(golden_dir, zymo_read, template_model):
    """test_continuousPairHmm_em (tests/signalPairwiseTest.c:1604-1714) through the host API: ten E/M rounds
    on the reference's Zymo template read from a random model (anchors taken from the un-banded posterior
    instead of lastz);
    the expected likelihood must not fall, by the reference's own criterion."""
    L = h.lib()
    sm = L.getStrawManStateMachine3(os.path.join(golden_dir, "template_median68pA.model").encode())
    L.emissions_signal_scaleModel(sm, *zymo_read["template_params"])
    rd = h.Read(zymo_read["reference"], zymo_read["template_events"])
    om = o.Sm3Model(template_model[0], template_model[2]).scaled(*zymo_read["template_params"])
    ref = o.aligned_pairs_without_banding(om, zymo_read["reference"], rd.lX, zymo_read["template_events"],
                                          o.default_params())
    best = {int(x): int(y) for q, x, y in ref["triples"] if q > 9000000}
    anchors, py = [], -1
    for x in sorted(best)[::25]:
        if best[x] > py:
            anchors.append((x, best[x]))
            py = best[x]
    lst = h.make_anchor_list(anchors)
    p = L.pairwiseAlignmentBandingParameters_construct()
    p.contents.minDiagsBetweenTraceBack = 300
    # start from a random model, as the reference does (continuousPairHmm_randomize :193-204)
    rng = np.random.default_rng(17)
    e = h.Expectations()
    for i in range(9):
        e.transitions[i] = rng.random()
    for i in range(h.NUM_KMERS):
        e.individualKmerGapProbs[i] = rng.random()
    L.cpecan_pairHmmExpectations_normalize(C.byref(e))
    L.cpecan_pairHmmExpectations_load(sm, C.byref(e))
    prev, first = -np.inf, None
    for it in range(10):
        e = h.Expectations()
        L.cpecan_getSignalExpectationsUsingAnchors(sm, C.byref(e), rd.sX, rd.sY, lst, p, False, False)   # E step
        assert np.isfinite(e.likelihood)
        assert prev <= e.likelihood * 0.95
        prev = e.likelihood
        first = first if first is not None else e.likelihood
        L.cpecan_pairHmmExpectations_normalize(C.byref(e))
        L.cpecan_pairHmmExpectations_load(sm, C.byref(e))                      # M step
    assert prev > first  # ten rounds fit the read better than the random start did
    L.stList_destruct(lst)
    L.pairwiseAlignmentBandingParameters_destruct(p)
    L.stateMachine_destruct(sm)


@pytest.mark.gpu
def test_hdp_machine_through_host_api(golden_dir, tmp_path):
    """deserialize_nhdp (the C reader of the host library) + getHdpStateMachine3 + sequence_getKmer3 +
    getAlignedPairsUsingAnchors on the reference's serialized HDP, against the oracle fed by the
    independent Python reader of the same file."""
    L = h.lib()
    path = os.path.join(golden_dir, "testTemplate.nhdp")
    nh = L.deserialize_nhdp(path.encode())
    assert L.get_nanopore_hdp_alphabet_size(nh) == 6
    sm = L.getHdpStateMachine3(nh)
    parsed = o.load_nhdp(path)
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
    ev = np.ascontiguousarray(np.array(ev).reshape(-1))
    xbuf = C.create_string_buffer(x.encode())
    sX = L.sequence_construct2(lX, C.cast(xbuf, C.c_void_p), h.fn_ptr("sequence_getKmer3"),
                               h.fn_ptr("sequence_sliceNucleotideSequence2"))
    sY = L.sequence_construct2(ev.size // 3, ev.ctypes.data_as(C.c_void_p), h.fn_ptr("sequence_getEvent"),
                               h.fn_ptr("sequence_sliceEventSequence2"))
    p = L.pairwiseAlignmentBandingParameters_construct()
    p.contents.minDiagsBetweenTraceBack = 100
    lst = h.make_anchor_list(anchors)
    pairs = L.getAlignedPairsUsingAnchors(sm, sX, sY, lst, p, h.fn_ptr("diagonalCalculationPosteriorMatchProbs"),
                                          True, True)
    got = h.list_to_array(pairs)
    L.stList_destruct(pairs)
    ref = o.aligned_pairs_using_anchors(om, x, lX, ev.reshape(-1, 3), anchors,
                                        o.default_params(minDiagsBetweenTraceBack=100), True, True)
    assert len(got) > lX // 2
    assert np.array_equal(got, ref["triples"])  # same pairs, same order, same integer posteriors

    # the E-step of the same alignment: transitions, likelihood, assignments; then the .expectations file
    hmm = L.cpecan_hdpExpectations_construct(0.0, 0.05)
    L.cpecan_getHdpExpectationsUsingAnchors(sm, hmm, sX, sY, lst, p, True, True)
    want = o.expectations_h_using_anchors(om, [(x, lX, ev.reshape(-1, 3), anchors)],
                                          o.default_params(minDiagsBetweenTraceBack=100), 0.05, True, True)
    e = hmm.contents
    assert np.allclose(list(e.transitions), want["transitions"], rtol=1e-9, atol=1e-12)
    assert np.isclose(e.likelihood, want["likelihood"], rtol=1e-12)
    n = e.numberOfAssignments
    assert n == len(want["assign"]) and n > 20
    kmers = [e.kmerAssignments[i * 7:i * 7 + 6].decode() for i in range(n)]
    assert kmers == [x[int(ix):int(ix) + 6] for _, ix, _ in want["assign"]]
    assert [e.eventAssignments[i] for i in range(n)] == [ev[3 * int(iy)] for _, _, iy in want["assign"]]
    path = os.path.join(str(tmp_path), "t.expectations")
    libc, fh = _c_file(path, b"w")
    L.cpecan_hdpExpectations_write(hmm, fh)
    libc.fclose(fh)
    lines = open(path).read().split("\n")
    assert lines[0].split() == ["7", "3", "0.050000", str(n)]  # type threeStateHdp, states, threshold, count
    assert len(lines[1].split()) == 10 and len(lines[2].split()) == n and lines[3].split() == kmers
    back = L.cpecan_hdpExpectations_read(path.encode()).contents  # the reader of the same file
    assert back.numberOfAssignments == n and back.threshold == 0.05
    assert [back.kmerAssignments[i * 7:i * 7 + 6].decode() for i in range(n)] == kmers
    assert np.allclose([back.eventAssignments[i] for i in range(n)], [e.eventAssignments[i] for i in range(n)],
                       atol=1e-6)  # "%lf" keeps six decimals
    assert np.allclose(list(back.transitions), list(e.transitions), atol=1e-6)
    L.cpecan_hdpExpectations_load(sm, hmm)  # un-normalised counts here: only the wiring is checked
    s3 = C.cast(sm, C.POINTER(h.StateMachine3)).contents
    assert s3.TRANSITION_MATCH_CONTINUE == np.log(e.transitions[0]) and s3.TRANSITION_GAP_SWITCH_TO_Y == -np.inf
    L.cpecan_hdpExpectations_destruct(hmm)
    L.stList_destruct(lst)
    L.sequence_sequenceDestroy(sX)
    L.sequence_sequenceDestroy(sY)
    L.pairwiseAlignmentBandingParameters_destruct(p)
    L.stateMachine_destruct(sm)
    L.destroy_nanopore_hdp(nh)


def test_npread_loader_remap_and_descale(golden_dir, zymo_read):
    """the .npRead reader and anchor re-mapping the reference's signal tests start from
    (tests/signalPairwiseTest.c:1007-1105, tests/nanoporeTest.c), on the reference's own read file"""
    L = h.lib()
    r = L.nanopore_loadNanoporeReadFromFile(os.path.join(golden_dir, "ZymoC_ch_1_file1.npRead").encode())
    n = r.contents
    assert (n.readLength, n.nbTemplateEvents, n.nbComplementEvents) == (
        zymo_read["read_length"], zymo_read["n_template"], zymo_read["n_complement"])
    assert n.readLength == len(n.twoDread) and n.twoDread.decode() == zymo_read["read"] and n.scaled
    tp, cpar = n.templateParams, n.complementParams
    assert [tp.scale, tp.shift, tp.var, tp.scale_sd, tp.var_sd] == zymo_read["template_params"]
    assert [cpar.scale, cpar.shift, cpar.var, cpar.scale_sd, cpar.var_sd] == zymo_read["complement_params"]
    tmap = np.ctypeslib.as_array(n.templateEventMap, shape=(n.readLength,))
    cmap = np.ctypeslib.as_array(n.complementEventMap, shape=(n.readLength,))
    tev = np.ctypeslib.as_array(n.templateEvents, shape=(3 * n.nbTemplateEvents,))
    cev = np.ctypeslib.as_array(n.complementEvents, shape=(3 * n.nbComplementEvents,))
    assert np.array_equal(tmap, zymo_read["template_map"]) and np.array_equal(cmap, zymo_read["complement_map"])
    assert np.array_equal(tev, zymo_read["template_events"])
    assert np.array_equal(cev, zymo_read["complement_events"])
    # remapping: (x, y in read coordinates) -> (x, event index), optionally re-based
    pairs = [(3, 0), (10, 7), (200, n.readLength - 1)]
    lst = h.make_anchor_list(pairs)
    m = L.nanopore_remapAnchorPairs(lst, n.templateEventMap)
    assert h.list_to_array(m, 2).tolist() == [[x, int(tmap[y])] for x, y in pairs]
    mo = L.nanopore_remapAnchorPairsWithOffset(lst, n.templateEventMap, 5)
    assert h.list_to_array(mo, 2).tolist() == [[x, int(tmap[y] - tmap[5])] for x, y in pairs]
    for q in (m, mo, lst):
        L.stList_destruct(q)
    # de-scaling keeps the reference's stride: every third slot below nbEvents
    want_t, want_c = zymo_read["template_events"].copy(), zymo_read["complement_events"].copy()
    it, ic = np.arange(0, n.nbTemplateEvents, 3), np.arange(0, n.nbComplementEvents, 3)
    want_t[it] = (want_t[it] - tp.shift) / tp.scale
    want_c[ic] = (want_c[ic] - cpar.shift) / cpar.scale
    L.nanopore_descaleNanoporeRead(r)
    assert not n.scaled and np.array_equal(tev, want_t) and np.array_equal(cev, want_c)
    L.nanopore_nanoporeReadDestruct(r)


def _c_file(path, mode):
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    return libc, libc.fopen(str(path).encode(), mode)


@pytest.mark.parametrize("symbols,hmm_type", [(4, 0), (5, 1)])
def test_hmm_discrete_file_round_trip_and_normalisation(tmp_path, symbols, hmm_type):
    """tests/pairwiseAlignerTest.c:754-855 (test_hmmDiscrete for fiveState / fiveStateAsymmetric): counts in,
    written, re-read identical; hmmDiscrete_normalize2's known answers"""
    L = h.lib()
    hmm = h.new_hmm_discrete(0.0, symbols, hmm_type)
    for f in range(5):
        for t in range(5):
            L.hmmDiscrete_addToTransitionExpectation(hmm, f, t, float(f * 5 + t))
    for st in range(5):
        for x in range(symbols):
            for y in range(symbols):
                L.hmmDiscrete_addToEmissionExpectation(hmm, st, x, y, float(st * symbols * symbols + x * symbols + y))
    path = tmp_path / "temp.hmm"
    libc, fh = _c_file(path, b"w")
    L.hmmDiscrete_write.argtypes = [C.POINTER(h.HmmDiscrete), C.c_void_p]
    L.hmmDiscrete_write(hmm, fh)
    libc.fclose(fh)
    L.hmmDiscrete_destruct(hmm)
    lines = open(path).read().split("\n")
    assert lines[0].split() == [str(hmm_type), "5", str(symbols)]
    assert len(lines[1].split()) == 26 and len(lines[2].split()) == 5 * symbols * symbols
    assert lines[1].split()[7] == "7.000000"  # "%f" fields (quirk Q7)
    hmm = L.hmmDiscrete_loadFromFile(str(path).encode())
    b = hmm.contents.baseHmm
    assert (b.type, b.stateNumber, b.symbolSetSize, b.matrixSize) == (hmm_type, 5, symbols, symbols * symbols)
    for f in range(5):
        for t in range(5):
            assert L.hmmDiscrete_getTransitionExpectation(hmm, f, t) == f * 5 + t
    for st in range(5):
        for x in range(symbols):
            for y in range(symbols):
                assert L.hmmDiscrete_getEmissionExpectation(hmm, st, x, y) == st * symbols * symbols + x * symbols + y
    L.hmmDiscrete_normalize2(hmm, True)
    m = symbols * symbols
    for f in range(5):
        z = f * 25 + 10
        for t in range(5):
            assert L.hmmDiscrete_getTransitionExpectation(hmm, f, t) == (f * 5 + t) / z
    for st in range(5):
        total = m * m * st + (m * (m - 1)) // 2
        for x in range(symbols):
            for y in range(symbols):
                assert L.hmmDiscrete_getEmissionExpectation(hmm, st, x, y) == (st * m + x * symbols + y) / total
    L.hmmDiscrete_destruct(hmm)
    assert L.emissions_discrete_getBaseIndex(b"G") == 2 and L.emissions_discrete_getBaseIndex(b"n") == 4097


def test_continuous_pair_hmm_file_round_trip(tmp_path):
    """tests/signalPairwiseTest.c:1461-1542: transitions, likelihood and k-mer gap counts through a .hmm file"""
    L = h.lib()
    e = h.Expectations()
    for i in range(9):
        e.transitions[i] = float(i)
    for i in range(h.NUM_KMERS):
        e.individualKmerGapProbs[i] = float(4096 * 3 + i)
    e.likelihood = -1234.5
    path = tmp_path / "temp.hmm"
    libc, fh = _c_file(path, b"w")
    L.cpecan_pairHmmExpectations_write.argtypes = [C.POINTER(h.Expectations), C.c_void_p]
    L.cpecan_pairHmmExpectations_write(C.byref(e), fh)
    libc.fclose(fh)
    lines = open(path).read().split("\n")
    assert lines[0].split() == ["2", "3", "4096"] and len(lines[1].split()) == 10 and len(lines[2].split()) == 4096
    r = L.cpecan_pairHmmExpectations_read(str(path).encode()).contents
    assert list(r.transitions) == list(e.transitions) and r.likelihood == e.likelihood
    assert list(r.individualKmerGapProbs) == list(e.individualKmerGapProbs)
    L.cpecan_pairHmmExpectations_normalize(C.byref(r))
    assert r.transitions[4] == 4.0 / 12.0 and abs(sum(r.individualKmerGapProbs) - 1.0) < 1e-12


def test_vanilla_hmm_file_round_trip(golden_dir, tmp_path):
    """tests/signalPairwiseTest.c:1544-1602 (test_vanillaHmm): skip-bin counts and the two emission tables of the
    state machine through a .hmm file (tables to the file's "%f" precision, as the reference checks: 1e-3), then
    vanillaHmm_normalizeKmerSkipBins' known answer"""
    L = h.lib()
    model = os.path.join(golden_dir, "template_median68pA.model").encode()
    sm = L.getSignalStateMachine3Vanilla(model)
    hmm = h.VanillaExpectations()
    for i in range(60):
        hmm.kmerSkipBins[i] = float(4096 * 3 + i)
    hmm.likelihood = -77.25
    path = tmp_path / "v.hmm"
    libc, fh = _c_file(path, b"w")
    L.cpecan_vanillaExpectations_write(C.byref(hmm), sm, fh)
    libc.fclose(fh)
    lines = open(path).read().split("\n")
    assert lines[0].split() == ["4", "3", "4096"] and len(lines[1].split()) == 61
    assert len(lines[2].split()) == len(lines[3].split()) == 1 + 4096 * 5
    blank = L.getSignalStateMachine3Vanilla(model)
    b = C.cast(blank, C.POINTER(h.StateMachine)).contents
    a = C.cast(sm, C.POINTER(h.StateMachine)).contents
    n = 1 + 4096 * 5
    np.ctypeslib.as_array(b.EMISSION_MATCH_PROBS, shape=(n,))[:] = 0.0
    np.ctypeslib.as_array(b.EMISSION_GAP_Y_PROBS, shape=(n,))[:] = 0.0
    back = L.cpecan_vanillaExpectations_read(str(path).encode(), blank).contents
    assert list(back.kmerSkipBins) == list(hmm.kmerSkipBins) and back.likelihood == hmm.likelihood
    for t in ("EMISSION_MATCH_PROBS", "EMISSION_GAP_Y_PROBS"):
        assert np.allclose(np.ctypeslib.as_array(getattr(b, t), shape=(n,)),
                           np.ctypeslib.as_array(getattr(a, t), shape=(n,)), atol=1e-3)
    L.cpecan_vanillaExpectations_normalize(C.byref(back))
    total = sum(4096 * 3 + i for i in range(60))
    assert all(back.kmerSkipBins[i] == (4096 * 3 + i) / total for i in range(60))
    L.stateMachine_destruct(sm)
    L.stateMachine_destruct(blank)


def _sm5_as_oracle_model(sm):
    """an oracle 5-state model carrying the transitions and emissions of a host StateMachine5"""
    m = o.Sm5Model()
    s = sm.contents
    for i, (name, _) in enumerate(h.StateMachine5._fields_[1:]):
        m.c.t[i] = getattr(s, name)
    m.match[:] = np.ctypeslib.as_array(s.model.EMISSION_MATCH_PROBS, shape=(16,))
    m.gx[:] = np.ctypeslib.as_array(s.model.EMISSION_GAP_X_PROBS, shape=(4,))
    m.gy[:] = np.ctypeslib.as_array(s.model.EMISSION_GAP_Y_PROBS, shape=(4,))
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("hmm_type", [0, 1])
def test_discrete_baum_welch_from_a_random_model(hmm_type):
    """tests/pairwiseAlignerTest.c:857-945 (test_HmmDiscrete_em): start from hmmDiscrete_randomize, ten
    rounds of E-step (getExpectationsUsingAnchors on the GPU) / hmmDiscrete_normalize2 / M-step
    (getStateMachine5); the likelihood must not fall by more than the reference's 5 % allowance, and every
    E-step is compared with the oracle run on the machine of that round."""
    L = h.lib()
    rng = np.random.default_rng(19 + hmm_type)
    for trial in range(3):
        x = "".join(rng.choice(list("ACGT"), int(rng.integers(30, 100))))
        y = "".join(ch if rng.random() > 0.2 else rng.choice(list("ACGT")) for ch in x if rng.random() > 0.05)
        xb, yb = C.create_string_buffer(x.encode()), C.create_string_buffer(y.encode())
        sX = L.sequence_construct2(len(x), C.cast(xb, C.c_void_p), h.fn_ptr("sequence_getBase"),
                                   h.fn_ptr("sequence_sliceNucleotideSequence2"))
        sY = L.sequence_construct2(len(y), C.cast(yb, C.c_void_p), h.fn_ptr("sequence_getBase"),
                                   h.fn_ptr("sequence_sliceNucleotideSequence2"))
        p = L.pairwiseAlignmentBandingParameters_construct()
        op = o.default_params()
        fns = L.stateMachineFunctions_construct(h.fn_ptr("emissions_symbol_getGapProb"),
                                                h.fn_ptr("emissions_symbol_getGapProb"),
                                                h.fn_ptr("emissions_symbol_getMatchProb"))
        hmm = h.new_hmm_discrete(0.0, 4, hmm_type)
        L.hmmDiscrete_randomize(hmm)
        sm = L.getStateMachine5(hmm, fns)
        L.hmmDiscrete_destruct(hmm)
        lst = h.make_anchor_list([])
        previous = -np.inf
        for it in range(10):
            hmm = h.new_hmm_discrete(1e-12, 4, hmm_type)
            L.getExpectationsUsingAnchors(sm, hmm, sX, sY, lst, p, h.fn_ptr("diagonalCalculation_Expectations"),
                                          False, False)
            ref = o.expectations5_using_anchors(_sm5_as_oracle_model(sm), x, len(x), y, np.zeros((0, 2), np.int64),
                                                op, o.OrcExpectations5()).as_array()
            got = np.concatenate([np.ctypeslib.as_array(hmm.contents.transitions, shape=(25,)),
                                  np.ctypeslib.as_array(hmm.contents.emissions, shape=(80,)),
                                  [hmm.contents.baseHmm.likelihood]])
            assert np.allclose(got, ref + np.r_[np.full(105, 1e-12), 0.0], rtol=1e-9, atol=1e-12)
            L.hmmDiscrete_normalize2(hmm, True)
            like = hmm.contents.baseHmm.likelihood
            assert previous <= like * 0.95
            previous = like
            L.stateMachine_destruct(sm)
            sm = L.getStateMachine5(hmm, fns)
            L.hmmDiscrete_destruct(hmm)
        assert np.isfinite(previous)
        L.stateMachine_destruct(sm)
        L.stList_destruct(lst)
        L.sequence_sequenceDestroy(sX)
        L.sequence_sequenceDestroy(sY)
        L.pairwiseAlignmentBandingParameters_destruct(p)


def test_reweighting_by_gap_probability():
    """getIndelProbabilities / reweightAlignedPairs2 (impl/pairwiseAligner.c:1619-1667) against their
    definition: gap mass per position = 1e7 - sum of its pair weights (floored at 0); new weight =
    weight - gapGamma * (gapX[x] + gapY[y]) with the reference's int64 - double -> int64 conversion"""
    L = h.lib()
    L.stIntTuple_construct3.restype = C.c_void_p
    L.stIntTuple_construct3.argtypes = [C.c_int64] * 3
    tri = [(9000000, 0, 0), (600000, 0, 1), (7000000, 1, 1), (5000000, 2, 3), (6000000, 2, 3), (123457, 3, 2)]
    lst = L.stList_construct3(0, h.fn_ptr("stIntTuple_destruct"))
    for t in tri:
        L.stList_append(lst, L.stIntTuple_construct3(*t))
    gx = L.getIndelProbabilities(lst, 4, True)
    gy = L.getIndelProbabilities(lst, 4, False)
    want_x, want_y = [10000000] * 4, [10000000] * 4
    for w, x, y in tri:
        want_x[x] -= w
        want_y[y] -= w
    want_x, want_y = [max(v, 0) for v in want_x], [max(v, 0) for v in want_y]
    assert [gx[i] for i in range(4)] == want_x and [gy[i] for i in range(4)] == want_y
    assert want_x[2] == 0  # over-full position is floored
    assert L.reweightAlignedPairs2(lst, 4, 4, 0.0) == lst  # gapGamma <= 0: the list itself
    out = L.reweightAlignedPairs2(lst, 4, 4, 0.3)
    got = h.list_to_array(out)
    want = [[int(w - 0.3 * (want_x[x] + want_y[y])), x, y] for w, x, y in tri]
    assert got.tolist() == want
    L.stList_destruct(out)


def test_tsv_writer_reproduces_the_references_own_output(golden_dir, zymo_read, tmp_path):
    """writePosteriorProbs (vanillaAlign.c:26-96) against tests/test_alignments/simple_alignment.tsv, the
    reference's own output for the shipped Zymo read (template and complement strands): the aligned pairs,
    event indices and posteriors are read back from that file, everything else -- reference positions, both
    k-mer columns, event observations, scaled model levels, de-scaled means, the number formats -- is produced
    here from the read, the two pore models and the reference sequence, and must match byte for byte."""
    L = h.lib()
    want = open(os.path.join(golden_dir, "simple_alignment.tsv")).read().split("\n")
    want = [w for w in want if w]
    ref = zymo_read["reference"]
    rc = "".join({"A": "T", "C": "G", "G": "C", "T": "A"}[c] for c in reversed(ref))
    out = tmp_path / "out.tsv"
    L.stIntTuple_construct3.restype = C.c_void_p
    L.stIntTuple_construct3.argtypes = [C.c_int64] * 3
    read_file = want[0].split("\t")[3]
    for strand, label, model_file, params, events, target, ref_offset in (
            (0, "t", "template_median68pA.model", zymo_read["template_params"], zymo_read["template_events"], ref, 0),
            (1, "c", "complement_median68pA_pop2.model", zymo_read["complement_params"],
             zymo_read["complement_events"], rc, len(ref))):
        sm = L.getStrawManStateMachine3(os.path.join(golden_dir, model_file).encode())
        L.emissions_signal_scaleModel(sm, *params)
        rows = [w.split("\t") for w in want if w.split("\t")[4] == label]
        lst = L.stList_construct3(0, h.fn_ptr("stIntTuple_destruct"))
        for r in rows:
            x_adj, y, p = int(r[1]), int(r[5]), float(r[12])
            x = x_adj if strand == 0 else (len(ref) - 6) - x_adj - (len(ref) - ref_offset)
            L.stList_append(lst, L.stIntTuple_construct3(int(round(p * 1e7)), x, y))
        ev = np.ascontiguousarray(events, dtype=np.float64).reshape(-1)
        L.writePosteriorProbs(str(out).encode(), read_file.encode(), sm.contents.model.EMISSION_MATCH_PROBS,
                              params[0], params[1], ev.ctypes.data_as(C.POINTER(C.c_double)), target.encode(), True,
                              b"ZYMO", 0, ref_offset, lst, strand)
        L.stList_destruct(lst)
        L.stateMachine_destruct(sm)
    got = [g for g in open(out).read().split("\n") if g]
    # the reference interleaves nothing: template rows then complement rows, as vanillaAlign writes them
    assert len(got) == len(want) == 1907
    assert got == [w for w in want if w.split("\t")[4] == "t"] + [w for w in want if w.split("\t")[4] == "c"]


def test_diagonal_band_iterator_logadd_and_overlap_filter():
    """the geometry and utility functions the reference exports and tests (tests/pairwiseAlignerTest.c:
    test_diagonal :22, test_bands :74, test_logAdd :139, test_filterToRemoveOverlap :515), host-only"""
    L = h.lib()
    xL, yL, xU, yU = 10, 20, 30, 0
    d = L.diagonal_construct(xL + yL, xL - yL, xU - yU)
    assert L.diagonal_getXay(d) == xL + yL and L.diagonal_getMinXmy(d) == xL - yL
    assert L.diagonal_getMaxXmy(d) == xU - yU and L.diagonal_getWidth(d) == (xU - yU - (xL - yL)) // 2 + 1
    assert L.diagonal_getXCoordinate(xL + yL, xL - yL) == xL and L.diagonal_getYCoordinate(xL + yL, xL - yL) == yL
    assert L.diagonal_equals(d, d) and not L.diagonal_equals(d, L.diagonal_construct(0, 0, 0))

    # test_bands: anchors (1,0) (2,1) (3,3), lX 7, lY 5, expansion 2, and the iterator's clamping at both ends
    anchors = [(1, 0), (2, 1), (3, 3)]
    lst = h.make_anchor_list(anchors)
    band = L.band_construct(lst, 7, 5, 2)
    lo, hi = o.band(anchors, 7, 5, 2)
    it = L.bandIterator_construct(band)
    for k in range(13):
        dd = L.bandIterator_getNext(it)
        assert (dd.xay, dd.xmyL, dd.xmyR) == (k, lo[k], hi[k])
    for _ in range(3):  # past the end: the last diagonal again
        dd = L.bandIterator_getNext(it)
        assert (dd.xay, dd.xmyL, dd.xmyR) == (12, lo[12], hi[12])
    clone = L.bandIterator_clone(it)
    for k in range(12, -1, -1):
        dd = L.bandIterator_getPrevious(it)
        assert (dd.xay, dd.xmyL, dd.xmyR) == (k, lo[k], hi[k])
    dd = L.bandIterator_getPrevious(it)
    assert dd.xay == 0
    assert L.bandIterator_getPrevious(clone).xay == 12
    L.bandIterator_destruct(it)
    L.bandIterator_destruct(clone)
    L.band_destruct(band)
    L.stList_destruct(lst)

    rng = np.random.default_rng(3)
    for _ in range(20000):
        i, j = rng.random(), rng.random()
        got = L.logAdd(np.log(i), np.log(j))
        assert abs(np.exp(got) - (i + j)) < 0.001
        assert got == o.lib().orc_logAdd(np.log(i), np.log(j))  # bit-identical to the oracle's
    assert L.logAdd(-np.inf, -3.0) == -3.0 and L.logAdd(-3.0, -np.inf) == -3.0

    for _ in range(10):
        lX, lY, acc = rng.integers(0, 60), rng.integers(0, 60), rng.random()
        pairs = [(x, y) for x in range(lX) for y in range(lY) if rng.random() > acc]
        lst = h.make_anchor_list(pairs)
        out = L.filterToRemoveOverlap(lst)
        got = {tuple(r) for r in h.list_to_array(out, 2)}
        arr = h.list_to_array(out, 2)
        assert all(arr[k, 0] < arr[k + 1, 0] and arr[k, 1] < arr[k + 1, 1] for k in range(len(arr) - 1))
        want = {(x, y) for (x, y) in pairs
                if not any((x2 <= x and y2 >= y) or (x2 >= x and y2 <= y) for (x2, y2) in pairs if (x2, y2) != (x, y))}
        assert got == want
        L.stList_destruct(out)
        L.stList_destruct(lst)


@pytest.mark.gpu
def test_single_banded_call_appends_in_emission_order(golden_dir):
    """getPosteriorProbsWithBanding (one call, no splitting) appends the triples as the reference's
    diagonal function does; getAlignedPairsUsingAnchors returns the same list tail first (:1447-1454)"""
    L = h.lib()
    rng = np.random.default_rng(9)
    x = "".join(rng.choice(list("ACGT"), 150))
    y = "".join(ch if rng.random() > 0.1 else rng.choice(list("ACGT")) for ch in x)
    sm = L.stateMachine5_construct(0, 4, h.fn_ptr("emissions_symbol_setEmissionsToDefaults"),
                                   h.fn_ptr("emissions_symbol_getGapProb"), h.fn_ptr("emissions_symbol_getGapProb"),
                                   h.fn_ptr("emissions_symbol_getMatchProb"), h.fn_ptr("cell_updateExpectations"))
    xb, yb = C.create_string_buffer(x.encode()), C.create_string_buffer(y.encode())
    sX = L.sequence_construct2(len(x), C.cast(xb, C.c_void_p), h.fn_ptr("sequence_getBase"),
                               h.fn_ptr("sequence_sliceNucleotideSequence"))
    sY = L.sequence_construct2(len(y), C.cast(yb, C.c_void_p), h.fn_ptr("sequence_getBase"),
                               h.fn_ptr("sequence_sliceNucleotideSequence"))
    p = L.pairwiseAlignmentBandingParameters_construct()
    p.contents.minDiagsBetweenTraceBack = 60
    p.contents.traceBackDiagonals = 10
    lst = h.make_anchor_list([(30, 30), (90, 90)])
    dest = L.stList_construct3(0, h.fn_ptr("stIntTuple_destruct"))
    extra = (C.c_void_p * 1)(dest)
    L.getPosteriorProbsWithBanding(sm, lst, sX, sY, p, False, False,
                                   h.fn_ptr("diagonalCalculationPosteriorMatchProbs"), C.cast(extra, C.c_void_p))
    raw = h.list_to_array(dest)
    pairs = L.getAlignedPairsUsingAnchors(sm, sX, sY, lst, p, h.fn_ptr("diagonalCalculationPosteriorMatchProbs"),
                                          False, False)
    got = h.list_to_array(pairs)
    assert len(raw) > 100 and np.array_equal(raw[::-1], got)
    for q in (dest, pairs, lst):
        L.stList_destruct(q)
    L.sequence_sequenceDestroy(sX)
    L.sequence_sequenceDestroy(sY)
    L.pairwiseAlignmentBandingParameters_destruct(p)
    L.stateMachine_destruct(sm)


def test_names_the_reference_declares_and_this_library_now_exports():
    """hmmDiscrete_normalize (inc/discreteHmm.h:38), hdpHmm_loadFromFile2 (inc/continuousHmm.h:109),
    convertPairwiseForwardStrandAlignmentToAnchorPairs (inc/pairwiseAligner.h:109), the 4-state machine's
    constructors (inc/stateMachine.h:289, :378)"""
    L = h.lib()
    for name in ("hmmDiscrete_normalize", "hdpHmm_loadFromFile2", "convertPairwiseForwardStrandAlignmentToAnchorPairs",
                 "stateMachine4_construct", "getStateMachine4"):
        assert hasattr(L, name), name


def test_convert_pairwise_alignment_to_anchor_pairs():
    """impl/pairwiseAligner.c:1039-1063 over sonLib's record: match runs as (x, y) pairs, trimmed at both ends"""
    L = h.lib()

    class Op(C.Structure):
        _fields_ = [("opType", C.c_int64), ("length", C.c_int64), ("score", C.c_float)]

    class List(C.Structure):
        _fields_ = [("length", C.c_int64), ("maxLength", C.c_int64), ("list", C.POINTER(C.c_void_p)), ("destroy", C.c_void_p)]

    class PA(C.Structure):
        _fields_ = [("contig1", C.c_char_p), ("start1", C.c_int64), ("end1", C.c_int64), ("strand1", C.c_int64),
                    ("contig2", C.c_char_p), ("start2", C.c_int64), ("end2", C.c_int64), ("strand2", C.c_int64),
                    ("score", C.c_float), ("operationList", C.POINTER(List))]

    ops = [Op(2, 5, 0.0), Op(0, 2, 0.0), Op(2, 3, 0.0), Op(1, 1, 0.0), Op(2, 4, 0.0)]  # M5 X2 M3 Y1 M4
    arr = (C.c_void_p * len(ops))(*[C.cast(C.pointer(o_), C.c_void_p) for o_ in ops])
    lst = List(len(ops), len(ops), arr, None)
    # INDEL_X: X advances alone (j += length); INDEL_Y: Y advances alone
    pa = PA(b"a", 10, 10 + 5 + 2 + 3 + 4, 1, b"b", 20, 20 + 5 + 3 + 1 + 4, 1, 0.0, C.pointer(lst))
    L.convertPairwiseForwardStrandAlignmentToAnchorPairs.restype = C.c_void_p
    L.convertPairwiseForwardStrandAlignmentToAnchorPairs.argtypes = [C.c_void_p, C.c_int64]
    got = h.list_to_array(L.convertPairwiseForwardStrandAlignmentToAnchorPairs(C.byref(pa), 1), 2)
    want = [(10 + l, 20 + l) for l in range(1, 4)] + [(17 + l, 25 + l) for l in range(1, 2)] + \
           [(20 + l, 29 + l) for l in range(1, 3)]
    assert [tuple(r) for r in got.tolist()] == want

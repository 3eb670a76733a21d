"""cpecan_trainModels (include/cpecan_api.h): the training loop of scripts/trainModels.py:244-330 (signal machines)
and cPecanEm.py:107-209 (discrete machine) as one native call, for each of the four machines.  Checked against the
same loop driven call by call from here through the reference-shaped entry points (an empty Hmm, getExpectationsUsingAnchors
per read, the normalisation, the loaders) -- whose E-steps are compared with the oracle in test_host_api.py,
test_vanilla_gpu.py, test_hdp_gpu.py and test_dna5_gpu.py; against the invariance of the M-step under two identical
ranks; and with the sum over the ranks taken by RCCL (cpecan_em_comm_reduce of libcpecan_em.so, one rank)."""
import ctypes as C
import os

import numpy as np
import pytest

import host_api as h
import synth
from cpecan_load import ROOT

pytestmark = pytest.mark.gpu

THREE_STATE, VANILLA, FIVE_STATE, FIVE_STATE_ASYM, THREE_STATE_HDP = 2, 4, 0, 1, 7
REDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int64)


def _bind(L):
    vp = C.c_void_p
    L.cpecan_trainModels.restype = vp
    L.cpecan_trainModels.argtypes = [C.c_int64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                     C.POINTER(h.Params), C.c_bool, C.c_bool, C.c_int, C.c_int64, C.c_double,
                                     C.c_double, vp, vp, C.POINTER(C.c_double)]
    L.getExpectationsUsingAnchorsBatch.argtypes = [C.c_int64, C.POINTER(vp), vp, C.POINTER(vp), C.POINTER(vp),
                                                   C.POINTER(vp), C.POINTER(h.Params), C.c_bool, C.c_bool]
    L.hmmContinuous_normalize.argtypes = [vp, C.c_int]
    L.continuousPairHmm_loadTransitionsAndKmerGapProbs.argtypes = [vp, vp]
    L.vanillaHmm_loadKmerSkipBinExpectations.argtypes = [vp, vp]
    L.hdpHmm_loadTransitions.argtypes = [vp, vp]
    return L


def _hd(hmm):
    return C.cast(hmm, C.POINTER(h.HmmDiscrete))


def _arr(ptrs):
    return (C.c_void_p * len(ptrs))(*[C.cast(p, C.c_void_p) for p in ptrs])


def _hmm_values(L, hmm, hmm_type):
    """(likelihood, "transition" table, emission table) of an Hmm through the exported accessors"""
    base = C.cast(hmm, C.POINTER(h.Hmm)).contents
    get_t = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int64, C.c_int64)(base.getTransitionsExpFcn)
    if hmm_type == VANILLA:
        return base.likelihood, np.array([get_t(hmm, b, 0) for b in range(60)]), np.zeros(0)
    n = base.stateNumber
    t = np.array([get_t(hmm, f, to) for f in range(n) for to in range(n)])
    e = np.zeros(0)
    if hmm_type == THREE_STATE:
        get_e = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int64, C.c_int64, C.c_int64)(base.getEmissionExpFcn)
        e = np.array([get_e(hmm, 0, k, 0) for k in range(h.NUM_KMERS)])
    if hmm_type in (FIVE_STATE, FIVE_STATE_ASYM):
        get_e = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_int64, C.c_int64, C.c_int64)(base.getEmissionExpFcn)
        e = np.array([get_e(hmm, s, x, y) for s in range(5) for x in range(4) for y in range(4)])
    return base.likelihood, t, e


def _destroy(L, hmm, hmm_type):
    if hmm_type in (FIVE_STATE, FIVE_STATE_ASYM):
        L.hmmDiscrete_destruct(_hd(hmm))
    else:
        L.hmmContinuous_destruct(hmm, hmm_type)


def _step_by_step(L, hmm_type, machines, reads, lists, p, iterations, pseudocount, threshold, ragged):
    """the loop, one reference-shaped call at a time; returns (likelihoods, values of the last normalised Hmm)"""
    likes, last = [], None
    for _ in range(iterations):
        if hmm_type in (FIVE_STATE, FIVE_STATE_ASYM):
            hmm = C.cast(h.new_hmm_discrete(pseudocount, 4, hmm_type), C.c_void_p)
        else:
            hmm = C.c_void_p(L.hmmContinuous_getEmptyHmm(hmm_type, pseudocount, threshold))
        for sm, (sX, sY), lst in zip(machines, reads, lists):
            L.getExpectationsUsingAnchors(sm, hmm, sX, sY, lst, p, h.fn_ptr("diagonalCalculation_Expectations"),
                                          ragged, ragged)
        likes.append(C.cast(hmm, C.POINTER(h.Hmm)).contents.likelihood)
        if hmm_type in (FIVE_STATE, FIVE_STATE_ASYM):
            L.hmmDiscrete_normalize2(_hd(hmm), True)
        elif hmm_type == THREE_STATE_HDP:
            L.hmmDiscrete_normalize2(_hd(hmm), False)
        else:
            L.hmmContinuous_normalize(hmm, hmm_type)
        for sm in {C.cast(m, C.c_void_p).value: m for m in machines}.values():
            if hmm_type in (FIVE_STATE, FIVE_STATE_ASYM):
                fresh = L.getStateMachine5(_hd(hmm), _step_by_step.fns)
                # the discrete loaders only exist inside getStateMachine5: take the loaded values over
                C.memmove(C.byref(C.cast(sm, C.POINTER(h.StateMachine5)).contents, h.StateMachine5.MATCH_CONTINUE.offset),
                          C.byref(fresh.contents, h.StateMachine5.MATCH_CONTINUE.offset),
                          17 * 8)
                for name in ("EMISSION_MATCH_PROBS", "EMISSION_GAP_X_PROBS", "EMISSION_GAP_Y_PROBS"):
                    dst = getattr(C.cast(sm, C.POINTER(h.StateMachine5)).contents.model, name)
                    src = getattr(fresh.contents.model, name)
                    C.memmove(dst, src, 8 * (16 if name == "EMISSION_MATCH_PROBS" else 4))
                L.stateMachine_destruct(fresh)
            elif hmm_type == THREE_STATE:
                L.continuousPairHmm_loadTransitionsAndKmerGapProbs(sm, hmm)
            elif hmm_type == VANILLA:
                L.vanillaHmm_loadKmerSkipBinExpectations(sm, hmm)
            else:
                L.hdpHmm_loadTransitions(sm, hmm)
        if last is not None:
            _destroy(L, last[0], hmm_type)
        last = (hmm, _hmm_values(L, hmm, hmm_type))
    _destroy(L, last[0], hmm_type)
    return np.array(likes), last[1]


def _train(L, hmm_type, machines, reads, lists, p, iterations, pseudocount, threshold, ragged, reduce=None,
           reduce_arg=None):
    n = len(machines)
    likes = np.zeros(iterations)
    hmm = C.c_void_p(L.cpecan_trainModels(n, _arr(machines), _arr([r[0] for r in reads]), _arr([r[1] for r in reads]),
                                          _arr(lists), p, ragged, ragged, hmm_type, iterations, pseudocount, threshold,
                                          reduce, reduce_arg, likes.ctypes.data_as(C.POINTER(C.c_double))))
    vals = _hmm_values(L, hmm, hmm_type)
    _destroy(L, hmm, hmm_type)
    return likes, vals


def _signal_reads(template_model, n, lX, lY, seed):
    out = []
    for r in range(n):
        rd = synth.make_read(np.random.default_rng(seed + r), template_model[0], lX + 7 * r, lY + 11 * r, anchor_every=40)
        out.append(rd)
    return out


def _machine_state(sm, hmm_type):
    if hmm_type in (FIVE_STATE, FIVE_STATE_ASYM):
        s = C.cast(sm, C.POINTER(h.StateMachine5)).contents
        t = np.ctypeslib.as_array((C.c_double * 17).from_address(C.addressof(s) + h.StateMachine5.MATCH_CONTINUE.offset)).copy()
        m = np.ctypeslib.as_array(C.cast(s.model.EMISSION_MATCH_PROBS, C.POINTER(C.c_double)), (16,)).copy()
        return np.concatenate([t, m])
    s = C.cast(sm, C.POINTER(h.StateMachine3)).contents
    t = np.array([getattr(s, f) for f, _ in h.StateMachine3._fields_ if f.startswith("TRANSITION_")])
    if hmm_type == THREE_STATE_HDP:
        return t
    gx = np.ctypeslib.as_array(C.cast(s.model.EMISSION_GAP_X_PROBS, C.POINTER(C.c_double)),
                               (60 if hmm_type == VANILLA else h.NUM_KMERS if hmm_type == THREE_STATE else 1,)).copy()
    return np.concatenate([t, gx]) if hmm_type != VANILLA else gx


@pytest.mark.parametrize("hmm_type", [THREE_STATE, VANILLA])
def test_signal_machines_trained_natively(hmm_type, golden_dir, template_model):
    L = _bind(h.lib())
    model = os.path.join(golden_dir, "template_median68pA.model").encode()
    make = L.getStrawManStateMachine3 if hmm_type == THREE_STATE else L.getSignalStateMachine3Vanilla
    getter = "sequence_getKmer" if hmm_type == THREE_STATE else "sequence_getKmer2"
    rds = _signal_reads(template_model, 5, 300, 420, 900 + hmm_type)
    p = L.pairwiseAlignmentBandingParameters_construct()
    p.contents.minDiagsBetweenTraceBack = 200
    results = []
    for native in (False, True):
        machines, reads, lists, keep = [], [], [], []
        for rd in rds:
            sm = make(model)
            L.emissions_signal_scaleModel(sm, *rd["scale_params"])
            xbuf = C.create_string_buffer(rd["seq"])
            ev = np.ascontiguousarray(rd["events"], dtype=np.float64).reshape(-1)
            sX = L.sequence_construct2(len(rd["seq"]) - 5, C.cast(xbuf, C.c_void_p), h.fn_ptr(getter),
                                       h.fn_ptr("sequence_sliceNucleotideSequence2"))
            sY = L.sequence_construct2(ev.size // 3, ev.ctypes.data_as(C.c_void_p), h.fn_ptr("sequence_getEvent"),
                                       h.fn_ptr("sequence_sliceEventSequence2"))
            keep += [xbuf, ev]
            machines.append(sm)
            reads.append((sX, sY))
            lists.append(h.make_anchor_list([tuple(a) for a in rd["anchors"]]))
        run = _train if native else _step_by_step
        likes, vals = run(L, hmm_type, machines, reads, lists, p, 3, 0.001, 0.0, False)
        results.append((likes, vals, [_machine_state(sm, hmm_type) for sm in machines]))
        for sm, (sX, sY), lst in zip(machines, reads, lists):
            L.stateMachine_destruct(sm)
            L.sequence_sequenceDestroy(sX)
            L.sequence_sequenceDestroy(sY)
            L.stList_destruct(lst)
    (la, va, ma), (lb, vb, mb) = results
    assert np.all(np.isfinite(la))
    if hmm_type == THREE_STATE:
        assert la[2] > la[0]  # three rounds fit the reads better (the vanilla machine's bin update is no EM step)
    assert np.allclose(la, lb, rtol=1e-10)
    assert np.allclose(va[1], vb[1], rtol=1e-8, atol=1e-14) and np.allclose(va[2], vb[2], rtol=1e-8, atol=1e-14)
    for a, b in zip(ma, mb):  # every read's machine carries the last M-step
        assert np.allclose(a, b, rtol=1e-8, atol=1e-12, equal_nan=True)
    assert abs(va[1].sum() - (3.0 if hmm_type == THREE_STATE else 1.0)) < 1e-9
    L.pairwiseAlignmentBandingParameters_destruct(p)


@pytest.mark.parametrize("hmm_type", [FIVE_STATE, FIVE_STATE_ASYM])
def test_discrete_machine_trained_natively(hmm_type):
    L = _bind(h.lib())
    rng = np.random.default_rng(31 + hmm_type)
    fns = L.stateMachineFunctions_construct(h.fn_ptr("emissions_symbol_getGapProb"),
                                            h.fn_ptr("emissions_symbol_getGapProb"),
                                            h.fn_ptr("emissions_symbol_getMatchProb"))
    _step_by_step.fns = fns
    pairs = []
    for _ in range(4):
        x = "".join(rng.choice(list("ACGT"), int(rng.integers(60, 140))))
        y = "".join(ch if rng.random() > 0.2 else rng.choice(list("ACGT")) for ch in x if rng.random() > 0.05)
        pairs.append((x, y))
    p = L.pairwiseAlignmentBandingParameters_construct()
    start = h.new_hmm_discrete(0.0, 4, hmm_type)
    L.hmmDiscrete_randomize(start)
    results = []
    for native in (False, True):
        sm = L.getStateMachine5(start, fns)  # ONE machine shared by the reads, as cPecanEm.py trains it
        reads, lists, keep = [], [], []
        for x, y in pairs:
            xb, yb = C.create_string_buffer(x.encode()), C.create_string_buffer(y.encode())
            keep += [xb, yb]
            reads.append((L.sequence_construct2(len(x), C.cast(xb, C.c_void_p), h.fn_ptr("sequence_getBase"),
                                                h.fn_ptr("sequence_sliceNucleotideSequence2")),
                          L.sequence_construct2(len(y), C.cast(yb, C.c_void_p), h.fn_ptr("sequence_getBase"),
                                                h.fn_ptr("sequence_sliceNucleotideSequence2"))))
            lists.append(h.make_anchor_list([]))
        run = _train if native else _step_by_step
        likes, vals = run(L, hmm_type, [sm] * len(pairs), reads, lists, p, 4, 1e-6, 0.0, False)
        results.append((likes, vals, _machine_state(sm, hmm_type)))
        L.stateMachine_destruct(sm)
        for (sX, sY), lst in zip(reads, lists):
            L.sequence_sequenceDestroy(sX)
            L.sequence_sequenceDestroy(sY)
            L.stList_destruct(lst)
    (la, va, ma), (lb, vb, mb) = results
    assert np.all(np.isfinite(la)) and la[3] > la[0]
    assert np.allclose(la, lb, rtol=1e-10)
    assert np.allclose(va[1], vb[1], rtol=1e-8) and np.allclose(va[2], vb[2], rtol=1e-8)
    assert np.allclose(ma, mb, rtol=1e-8, atol=1e-12)
    L.hmmDiscrete_destruct(start)
    L.pairwiseAlignmentBandingParameters_destruct(p)


def test_hdp_machine_transitions_trained_natively(golden_dir):
    import pyoracle as o
    L = _bind(h.lib())
    path = os.path.join(golden_dir, "testTemplate.nhdp")
    parsed = o.load_nhdp(path)
    om = o.HdpModel(parsed)
    rng = np.random.default_rng(77)
    data = []
    for r in range(3):
        lX = 120 + 20 * r
        x = "".join(rng.choice(list("ACGT"), lX + 5))
        ev, anchors = [], []
        for k in range(lX):
            row = parsed["kmer_row"][om.kmer_id(x[k:k + 6])]
            mode = parsed["grid"][int(np.argmax(parsed["y"][row]))]
            if k % 40 == 20:
                anchors.append((k, len(ev)))
            for _ in range(1 if rng.random() < 0.6 else 2):
                ev.append((mode + rng.normal(0, 1.0), 1.0, 0.01))
        data.append((x, lX, np.ascontiguousarray(np.array(ev).reshape(-1)), anchors))
    p = L.pairwiseAlignmentBandingParameters_construct()
    p.contents.minDiagsBetweenTraceBack = 100
    results = []
    for native in (False, True):
        nh = L.deserialize_nhdp(path.encode())
        sm = L.getHdpStateMachine3(nh)
        reads, lists, keep = [], [], []
        for x, lX, ev, anchors in data:
            xbuf = C.create_string_buffer(x.encode())
            keep.append(xbuf)
            reads.append((L.sequence_construct2(lX, C.cast(xbuf, C.c_void_p), h.fn_ptr("sequence_getKmer3"),
                                                h.fn_ptr("sequence_sliceNucleotideSequence2")),
                          L.sequence_construct2(ev.size // 3, ev.ctypes.data_as(C.c_void_p),
                                                h.fn_ptr("sequence_getEvent"), h.fn_ptr("sequence_sliceEventSequence2"))))
            lists.append(h.make_anchor_list(anchors))
        run = _train if native else _step_by_step
        likes, vals = run(L, THREE_STATE_HDP, [sm] * len(data), reads, lists, p, 3, 0.001, 0.05, True)
        results.append((likes, vals, _machine_state(sm, THREE_STATE_HDP)[:9]))
        L.stateMachine_destruct(sm)
        L.destroy_nanopore_hdp(nh)
        for (sX, sY), lst in zip(reads, lists):
            L.sequence_sequenceDestroy(sX)
            L.sequence_sequenceDestroy(sY)
            L.stList_destruct(lst)
    (la, va, ma), (lb, vb, mb) = results
    assert np.all(np.isfinite(la))
    assert np.allclose(la, lb, rtol=1e-10) and np.allclose(va[1], vb[1], rtol=1e-8)
    assert np.allclose(ma, mb, rtol=1e-8, atol=1e-12)
    L.pairwiseAlignmentBandingParameters_destruct(p)


def test_sum_over_ranks_two_identical_ranks_and_rccl(golden_dir, template_model):
    """reduce = "a second rank with the same reads" doubles every expectation and the likelihood: the normalised model
    is the one rank's, the likelihood twice it.  Then the real thing: libcpecan_em.so's communicator (RCCL, one rank)."""
    L = _bind(h.lib())
    model = os.path.join(golden_dir, "template_median68pA.model").encode()
    rds = _signal_reads(template_model, 3, 260, 380, 1300)
    p = L.pairwiseAlignmentBandingParameters_construct()
    calls = []

    @REDUCE_FN
    def twice(arg, values, n):
        calls.append(int(n))
        for i in range(n):
            values[i] *= 2.0

    em = C.CDLL(os.path.join(ROOT, "cpecan-signal_amd", "libcpecan_em.so"))
    em.cpecan_em_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
    em.cpecan_em_comm_destroy.argtypes = [C.c_void_p]
    comm = C.c_void_p()
    assert em.cpecan_em_comm_create(0, 0, 1, None, C.byref(comm)) == 0
    out = {}
    for name, fn, arg in (("one", None, None), ("twice", C.cast(twice, C.c_void_p), None),
                          ("rccl", C.cast(em.cpecan_em_comm_reduce, C.c_void_p), comm)):
        machines, reads, lists, keep = [], [], [], []
        for rd in rds:
            sm = L.getStrawManStateMachine3(model)
            L.emissions_signal_scaleModel(sm, *rd["scale_params"])
            xbuf = C.create_string_buffer(rd["seq"])
            ev = np.ascontiguousarray(rd["events"], dtype=np.float64).reshape(-1)
            keep += [xbuf, ev]
            machines.append(sm)
            reads.append((L.sequence_construct2(len(rd["seq"]) - 5, C.cast(xbuf, C.c_void_p), h.fn_ptr("sequence_getKmer"),
                                                h.fn_ptr("sequence_sliceNucleotideSequence2")),
                          L.sequence_construct2(ev.size // 3, ev.ctypes.data_as(C.c_void_p), h.fn_ptr("sequence_getEvent"),
                                                h.fn_ptr("sequence_sliceEventSequence2"))))
            lists.append(h.make_anchor_list([tuple(a) for a in rd["anchors"]]))
        out[name] = _train(L, THREE_STATE, machines, reads, lists, p, 2, 0.001, 0.0, False, fn, arg)
        for sm, (sX, sY), lst in zip(machines, reads, lists):
            L.stateMachine_destruct(sm)
            L.sequence_sequenceDestroy(sX)
            L.sequence_sequenceDestroy(sY)
            L.stList_destruct(lst)
    assert calls == [1 + 9 + h.NUM_KMERS] * 2
    assert np.allclose(out["twice"][0], 2.0 * out["one"][0], rtol=1e-12)
    assert np.allclose(out["twice"][1][1], out["one"][1][1], rtol=1e-10)
    assert np.allclose(out["twice"][1][2], out["one"][1][2], rtol=1e-10)
    # (the E-step's sums are accumulated by atomics in an order that differs from run to run: last-bit differences)
    assert np.allclose(out["rccl"][0], out["one"][0], rtol=1e-12)
    assert np.allclose(out["rccl"][1][1], out["one"][1][1], rtol=1e-10)
    assert np.allclose(out["rccl"][1][2], out["one"][1][2], rtol=1e-10)
    em.cpecan_em_comm_destroy(comm)
    L.pairwiseAlignmentBandingParameters_destruct(p)

"""getBlastPairs / getBlastPairsForPairwiseAlignmentParameters (impl/pairwiseAligner.c:1065-1281): the host code around
the external lastz executable.  lastz itself is not in this image (and the reference's copy is never run), so a
stand-in (tests/fake_lastz.py) answers the pipe: what is checked is everything the library does around it -- the
command line and the two sequences it sends, CIGAR parsing with insertions and deletions, trimming, ordering, the
overlap filter, the un-masked second pass inside large gaps, the size cut-off.  Parity with the real lastz's anchors is
unpinned."""
import ctypes as C
import os

import numpy as np
import pytest

import host_api as h

FAKE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_lastz.py")


@pytest.fixture()
def lastz(monkeypatch, tmp_path):
    log = str(tmp_path / "calls.log")
    monkeypatch.setenv("CPECAN_LASTZ", FAKE)
    monkeypatch.setenv("FAKE_LASTZ_LOG", log)
    L = h.lib()
    L.getBlastPairs.restype = C.c_void_p
    L.getBlastPairs.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_bool]
    L.getBlastPairsForPairwiseAlignmentParameters.restype = C.c_void_p
    L.getBlastPairsForPairwiseAlignmentParameters.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(h.Params)]
    return L, log


def _pairs(L, lst):
    out = h.list_to_array(lst, 2)
    L.stList_destruct(lst)
    return [tuple(int(v) for v in r) for r in out]


def test_cigar_lines_with_gaps_trim_and_order(lastz, monkeypatch):
    L, log = lastz
    # sequence 1 = "a" (X, second triple), sequence 2 = "b" (Y, first triple); D advances X only, I advances Y only
    monkeypatch.setenv("FAKE_LASTZ_LINES",
                       "#comment\n"
                       "cigar: b 40 52 + a 30 44 + 900 M 5 D 2 M 4 I 0 M 3\n"
                       "cigar: b 2 12 + a 1 9 + 700 M 4 I 2 M 4\n")
    got = _pairs(L, L.getBlastPairs(b"ACGT" * 20, b"ACGT" * 20, 1, True))
    first = [(1 + l, 2 + l) for l in range(1, 3)] + [(5 + l, 8 + l) for l in range(1, 3)]
    second = [(30 + l, 40 + l) for l in range(1, 4)] + [(37 + l, 45 + l) for l in range(1, 3)] + [(41 + 1, 49 + 1)]
    assert got == first + second  # sorted by x + y
    assert _pairs(L, L.getBlastPairs(b"", b"ACGT", 0, True)) == []
    line = open(log).read().split("\n")[0].split("\t")
    assert line[0] == ("--hspthresh=1800 --chain --strand=plus --gapped --format=cigar --gap=100,100 "
                       "--ambiguous=iupac,100,100") and line[4] == "ab"


def test_a_line_whose_operations_do_not_add_up_is_refused(lastz, monkeypatch, tmp_path):
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); import host_api as h; L = h.lib(); "
            "L.getBlastPairs.restype = C.c_void_p; L.getBlastPairs.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_bool]; "
            "L.getBlastPairs(b'ACGTACGTAC', b'ACGTACGTAC', 0, True)" % os.path.dirname(FAKE))
    env = dict(os.environ, FAKE_LASTZ_LINES="cigar: b 0 10 + a 0 9 + 100 M 10\n", CPECAN_LASTZ=FAKE)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "do not add up" in r.stderr


def test_two_level_anchoring_masks_then_unmasks_inside_large_gaps(lastz):
    L, log = lastz
    rng = np.random.default_rng(5)
    core = "".join(rng.choice(list("ACGT"), 900))
    # a soft-masked (lower case) stretch in the middle: the top-level, masked pass cannot anchor there (the stand-in
    # compares characters as they are sent; Y keeps it in upper case), the second pass sends upper case and can
    x = core[:300] + core[300:600].lower() + core[600:]
    y = core[:300] + core[300:450] + "T" + core[450:600] + core[600:]
    p = L.pairwiseAlignmentBandingParameters_construct()
    p.contents.constraintDiagonalTrim = 3
    p.contents.anchorMatrixBiggerThanThis = 1000
    p.contents.repeatMaskMatrixBiggerThanThis = 200 * 200
    got = _pairs(L, L.getBlastPairsForPairwiseAlignmentParameters(x.encode(), y.encode(), p))
    top = [(i, i) for i in range(3, 297)] + [(i, i + 1) for i in range(603, 897)]
    # the gap between the top-level anchors (296, 296) and (603, 604) is sent as X[297:603] / Y[297:604]: two blocks of
    # 153, trimmed by 3 in the sub-problem's own coordinates, shifted back
    inner = [(i, i) for i in range(300, 447)] + [(i, i + 1) for i in range(453, 600)]
    assert got == sorted(top + inner)
    calls = [l.split("\t") for l in open(log).read().split("\n") if l]
    assert [(int(c[1]), int(c[2]), int(c[3])) for c in calls] == [(900, 901, 0), (306, 307, 1)]  # masked, then upper case
    # below the size cut-off nothing is run at all
    p.contents.anchorMatrixBiggerThanThis = 900 * 901
    assert _pairs(L, L.getBlastPairsForPairwiseAlignmentParameters(x.encode(), y.encode(), p)) == []
    assert len([l for l in open(log).read().split("\n") if l]) == 2
    L.pairwiseAlignmentBandingParameters_destruct(p)


@pytest.mark.gpu
def test_dna_realignment_end_to_end_with_the_librarys_own_anchor_function(lastz):
    """What cPecan's realign does with one pair of sequences: getAlignedPairs(sM5, x, y, ..., getAnchorPairFcn =
    getBlastPairsForPairwiseAlignmentParameters) -- anchors from the aligner behind the pipe (the stand-in here),
    filtered, then the banded 5-state forward-backward on the GPU -- against the oracle given the same anchors."""
    import pyoracle as o
    L, _ = lastz
    rng = np.random.default_rng(23)
    x = "".join(rng.choice(list("ACGT"), 3000))
    y = []
    for i, ch in enumerate(x):  # substitutions, short deletions and insertions
        r = rng.random()
        if r < 0.04:
            y.append(rng.choice(list("ACGT")))
        elif r < 0.05:
            continue
        elif r < 0.06:
            y.append(ch + rng.choice(list("ACGT")))
        else:
            y.append(ch)
    y = "".join(y)
    p = L.pairwiseAlignmentBandingParameters_construct()
    p.contents.anchorMatrixBiggerThanThis = 500 * 500
    anchors_list = L.getBlastPairsForPairwiseAlignmentParameters(x.encode(), y.encode(), p)
    anchors = h.list_to_array(anchors_list, 2)
    L.stList_destruct(anchors_list)
    assert len(anchors) > 500 and np.all(np.diff(anchors[:, 0]) > 0) and np.all(np.diff(anchors[:, 1]) > 0)
    sm = L.stateMachine5_construct(0, 4, h.fn_ptr("emissions_symbol_setEmissionsToDefaults"),
                                   h.fn_ptr("emissions_symbol_getGapProb"), h.fn_ptr("emissions_symbol_getGapProb"),
                                   h.fn_ptr("emissions_symbol_getMatchProb"), h.fn_ptr("cell_updateExpectations"))
    xb, yb = C.create_string_buffer(x.encode()), C.create_string_buffer(y.encode())
    L.getAlignedPairs.restype = C.c_void_p
    pairs = L.getAlignedPairs(sm, C.cast(xb, C.c_void_p), C.cast(yb, C.c_void_p), len(x), len(y), p,
                              h.fn_ptr("sequence_getBase"), h.fn_ptr("sequence_getBase"),
                              h.fn_ptr("getBlastPairsForPairwiseAlignmentParameters"), False, False)
    got = h.list_to_array(pairs)
    L.stList_destruct(pairs)
    ref = o.aligned_pairs_using_anchors(o.Sm5Model(), x, len(x), y, [tuple(int(v) for v in a) for a in anchors],
                                        o.default_params(), False, False)
    assert len(got) > 2500
    assert np.array_equal(got, ref["triples"])  # same pairs, same order, same integer posteriors
    L.pairwiseAlignmentBandingParameters_destruct(p)
    L.stateMachine_destruct(sm)

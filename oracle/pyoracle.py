"""ctypes binding of the CPU oracle (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product (cpecan-signal_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NUM_KMERS = 4096
MODEL_LEN = 1 + NUM_KMERS * 5
SM3, SM5, VANILLA, HDP = 0, 1, 2, 3


class OrcModel(C.Structure):
    _fields_ = [("kind", C.c_int32), ("stateNumber", C.c_int32), ("t", C.c_double * 17),
                ("match", C.c_void_p), ("gapX", C.c_void_p), ("gapY", C.c_void_p),
                ("hdpRow", C.c_void_p), ("hdpGrid", C.c_void_p), ("hdpY", C.c_void_p),
                ("hdpSlope", C.c_void_p), ("gridLength", C.c_int32), ("alphabetSize", C.c_int32),
                ("alphabet", C.c_char * 16)]


class OrcParams(C.Structure):
    _fields_ = [("threshold", C.c_double), ("minDiagsBetweenTraceBack", C.c_int64),
                ("traceBackDiagonals", C.c_int64), ("diagonalExpansion", C.c_int64),
                ("splitMatrixBiggerThanThis", C.c_int64)]


class OrcExpectations(C.Structure):
    _fields_ = [("transitions", C.c_double * 9), ("kmerGap", C.c_double * NUM_KMERS),
                ("likelihood", C.c_double)]


class OrcExpectations5(C.Structure):
    _fields_ = [("transitions", C.c_double * 25), ("emissions", C.c_double * 80), ("likelihood", C.c_double)]

    def as_array(self):
        return np.concatenate([np.array(self.transitions), np.array(self.emissions), [self.likelihood]])


class OrcExpectationsV(C.Structure):
    _fields_ = [("kmerSkipBins", C.c_double * 60), ("likelihood", C.c_double)]

    def as_array(self):
        return np.concatenate([np.array(self.kmerSkipBins), [self.likelihood]])


class OrcExpectationsH(C.Structure):
    _fields_ = [("transitions", C.c_double * 9), ("likelihood", C.c_double), ("threshold", C.c_double),
                ("n", C.c_int64), ("cap", C.c_int64), ("assign", C.POINTER(C.c_int64)),
                ("logp", C.POINTER(C.c_double))]


class OrcResult(C.Structure):
    _fields_ = [("n", C.c_int64), ("cap", C.c_int64), ("triples", C.POINTER(C.c_int64)),
                ("logp", C.POINTER(C.c_double)), ("nTotals", C.c_int64), ("capTotals", C.c_int64),
                ("totalsXay", C.POINTER(C.c_int64)), ("totals", C.POINTER(C.c_double)),
                ("cells", C.c_int64)]


def build(force=False):
    if os.environ.get("CPECAN_ORACLE_LIB"):  # e.g. a sanitizer build of the same source
        return os.environ["CPECAN_ORACLE_LIB"]
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "cpecan_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liborc.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_logAdd.restype = C.c_double
        L.orc_logAdd.argtypes = [C.c_double, C.c_double]
        L.orc_kmer_index.restype = C.c_int64
        L.orc_kmer_index.argtypes = [C.c_char_p]
        L.orc_logGaussPdf.restype = C.c_double
        L.orc_logGaussPdf.argtypes = [C.c_double] * 3
        L.orc_strawman_match.restype = C.c_double
        L.orc_strawman_match.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_kmer_gap.restype = C.c_double
        L.orc_kmer_gap.argtypes = [C.c_void_p, C.c_int64]
        L.orc_scale_model.argtypes = [C.c_void_p] + [C.c_double] * 5
        L.orc_band.restype = C.c_int
        L.orc_band.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                               C.c_void_p]
        L.orc_split_points.restype = C.c_int64
        L.orc_split_points.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                       C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.orc_result_new.restype = C.POINTER(OrcResult)
        L.orc_result_free.argtypes = [C.POINTER(OrcResult)]
        L.orc_defaults_sm3_nanopore.argtypes = [C.POINTER(OrcModel)]
        L.orc_defaults_sm5.argtypes = [C.POINTER(OrcModel), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_defaults_vanilla.argtypes = [C.POINTER(OrcModel)]
        L.orc_defaults_hdp.argtypes = [C.POINTER(OrcModel)]
        L.orc_defaults_sm4.argtypes = [C.POINTER(OrcModel)]
        L.orc_hdp_density.restype = C.c_double
        L.orc_hdp_density.argtypes = [C.POINTER(OrcModel), C.c_char_p, C.c_double]
        L.orc_hdp_kmer_id.restype = C.c_int64
        L.orc_hdp_kmer_id.argtypes = [C.POINTER(OrcModel), C.c_char_p]
        L.orc_params_default.argtypes = [C.POINTER(OrcParams)]
        L.orc_aligned_pairs_using_anchors.restype = C.c_int
        L.orc_aligned_pairs_using_anchors.argtypes = [
            C.POINTER(OrcModel), C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
            C.c_int64, C.POINTER(OrcParams), C.c_int, C.c_int, C.c_void_p, C.POINTER(OrcResult)]
        L.orc_aligned_pairs_without_banding.restype = C.c_int
        L.orc_aligned_pairs_without_banding.argtypes = [
            C.POINTER(OrcModel), C.c_char_p, C.c_int64, C.c_void_p, C.c_int64,
            C.POINTER(OrcParams), C.c_int, C.c_int, C.POINTER(OrcResult)]
        L.orc_banded_dump.restype = C.c_int
        L.orc_banded_dump.argtypes = [
            C.POINTER(OrcModel), C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
            C.c_int64, C.POINTER(OrcParams), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
            C.POINTER(OrcResult)]
        L.orc_expectations_normalize.argtypes = [C.POINTER(OrcExpectations)]
        L.orc_expectations5_normalize.argtypes = [C.POINTER(OrcExpectations5)]
        L.orc_expectations_h_new.restype = C.POINTER(OrcExpectationsH)
        L.orc_expectations_h_new.argtypes = [C.c_double]
        L.orc_expectations_h_free.argtypes = [C.POINTER(OrcExpectationsH)]
        L.orc_expectations_h_using_anchors.restype = C.c_int
        L.orc_expectations_h_using_anchors.argtypes = [
            C.POINTER(OrcModel), C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
            C.POINTER(OrcParams), C.c_int, C.c_int, C.POINTER(OrcExpectationsH)]
        L.orc_expectations_v_using_anchors.restype = C.c_int
        L.orc_expectations_v_using_anchors.argtypes = [
            C.POINTER(OrcModel), C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
            C.POINTER(OrcParams), C.c_int, C.c_int, C.POINTER(OrcExpectationsV)]
        L.orc_expectations5_using_anchors.restype = C.c_int
        L.orc_expectations5_using_anchors.argtypes = [
            C.POINTER(OrcModel), C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
            C.POINTER(OrcParams), C.c_int, C.c_int, C.POINTER(OrcExpectations5)]
        _LIB = L
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_params(**kw):
    p = OrcParams()
    lib().orc_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class Sm3Model:
    """strawMan 3-state signal model: keeps the numpy tables alive behind the C struct."""

    def __init__(self, match, gap_y, gap_x=None, transitions=None):
        self.match = np.ascontiguousarray(match, dtype=np.float64)
        self.gap_y = np.ascontiguousarray(gap_y, dtype=np.float64)
        if gap_x is None:  # stateMachine3_construct: log(0.1) for every k-mer (stateMachine.c:1506)
            gap_x = np.full(NUM_KMERS, -2.3025850929940455)
        self.gap_x = np.ascontiguousarray(gap_x, dtype=np.float64)
        assert self.match.size == MODEL_LEN and self.gap_y.size == MODEL_LEN
        self.c = OrcModel()
        lib().orc_defaults_sm3_nanopore(C.byref(self.c))
        if transitions is not None:
            for i, v in enumerate(transitions):
                self.c.t[i] = v
        self.c.match = self.match.ctypes.data
        self.c.gapX = self.gap_x.ctypes.data
        self.c.gapY = self.gap_y.ctypes.data

    @property
    def transitions(self):
        return np.array([self.c.t[i] for i in range(9)])

    def scaled(self, scale, shift, var, scale_sd, var_sd):
        m = self.match.copy()
        lib().orc_scale_model(_ptr(m), scale, shift, var, scale_sd, var_sd)
        return Sm3Model(m, self.gap_y, self.gap_x, self.transitions)


class Sm4Model:
    """4-state signal model (getStateMachine4, impl/stateMachine.c:1750-1759): the strawMan emissions, the k-mer gap
    table left at zero by emissions_signal_initEmissionsToZero (:374-386), the template-read transitions of
    stateMachine4_construct (:992-1011)."""

    def __init__(self, match, gap_y, gap_x=None, transitions=None):
        self.match = np.ascontiguousarray(match, dtype=np.float64)
        self.gap_y = np.ascontiguousarray(gap_y, dtype=np.float64)
        self.gap_x = np.ascontiguousarray(np.zeros(NUM_KMERS) if gap_x is None else gap_x, dtype=np.float64)
        assert self.match.size == MODEL_LEN and self.gap_y.size == MODEL_LEN and self.gap_x.size == NUM_KMERS
        self.c = OrcModel()
        lib().orc_defaults_sm4(C.byref(self.c))
        if transitions is not None:
            for i, v in enumerate(transitions):
                self.c.t[i] = v
        self.c.match = self.match.ctypes.data
        self.c.gapX = self.gap_x.ctypes.data
        self.c.gapY = self.gap_y.ctypes.data

    @property
    def transitions(self):
        return np.array([self.c.t[i] for i in range(11)])

    def scaled(self, scale, shift, var, scale_sd, var_sd):
        m = self.match.copy()
        lib().orc_scale_model(_ptr(m), scale, shift, var, scale_sd, var_sd)
        return Sm4Model(m, self.gap_y, self.gap_x, self.transitions)


class VanillaModel:
    """3-state vanilla signal model (getSignalStateMachine3Vanilla, stateMachine.c:1761): match and
    extra-event tables as loaded, the 30 skip-bin values of the model file's second line used as both
    beta (0..29) and alpha (30..59), as emissions_signal_loadPoreModel stores them (:284-297)."""

    def __init__(self, match, skip30, gap_y, m_to_y_not_x=None, e_to_e=None):
        self.match = np.ascontiguousarray(match, dtype=np.float64)
        self.gap_y = np.ascontiguousarray(gap_y, dtype=np.float64)
        skip30 = np.asarray(skip30, dtype=np.float64)
        self.skip = np.ascontiguousarray(np.concatenate([skip30, skip30]) if skip30.size == 30 else skip30)
        assert self.match.size == MODEL_LEN and self.gap_y.size == MODEL_LEN and self.skip.size == 60
        self.c = OrcModel()
        lib().orc_defaults_vanilla(C.byref(self.c))
        if m_to_y_not_x is not None:
            self.c.t[0] = m_to_y_not_x
        if e_to_e is not None:
            self.c.t[1] = e_to_e
        self.c.match = self.match.ctypes.data
        self.c.gapX = self.skip.ctypes.data
        self.c.gapY = self.gap_y.ctypes.data

    @property
    def scalars(self):
        return np.array([self.c.t[i] for i in range(5)])

    def scaled(self, scale, shift, var, scale_sd, var_sd):
        m = self.match.copy()
        lib().orc_scale_model(_ptr(m), scale, shift, var, scale_sd, var_sd)
        return VanillaModel(m, self.skip, self.gap_y, self.c.t[0], self.c.t[1])


def load_nhdp(path):
    """Parse a serialized NanoporeHDP (serialize_nhdp impl/nanopore_hdp.c:820-843 / serialize_hdp
    impl/hdp.c:2880-3007; read back by deserialize_hdp :3009-3273) as far as densities need it: the
    alphabet, the sampling grid, every Dirichlet process's parent and, for the observed ones, the
    posterior-predictive values and spline slopes on the grid.  Returns a dict with, per k-mer id, the
    table row of its nearest observed ancestor (the walk of dir_proc_density :2588-2590)."""
    with open(path) as f:
        lines = f.read().split("\n")
    alphabet_size, alphabet, kmer_length = int(lines[0]), lines[1].strip(), int(lines[2])
    i = 3
    splines, has_data, sample_gamma, num_dps = (bool(int(lines[i])), bool(int(lines[i + 1])),
                                                bool(int(lines[i + 2])), int(lines[i + 3]))
    i += 4
    assert splines and has_data, "densities need a finalized HDP with data"
    i += 2                      # data, dp ids
    i += 1                      # base parameters mu nu alpha beta
    g0, g1, gl = lines[i].split()
    grid_start, grid_stop, grid_length = float(g0), float(g1), int(gl)
    i += 1
    i += 1                      # gamma
    if sample_gamma:
        i += 4                  # gamma alpha, gamma beta, w, s
    parent = np.full(num_dps, -1, np.int64)
    for d in range(num_dps):
        tok = lines[i + d].split()
        if tok[0] != "-":
            parent[d] = int(tok[0])
    i += num_dps
    rows, row_of = [], np.full(num_dps, -1, np.int64)
    for d in range(num_dps):
        tok = lines[i + d].split()
        if tok:
            row_of[d] = len(rows)
            rows.append(np.array(tok, dtype=np.float64))
    i += num_dps
    slopes = [None] * len(rows)
    for d in range(num_dps):
        tok = lines[i + d].split()
        if tok:
            slopes[row_of[d]] = np.array(tok, dtype=np.float64)
    n_kmers = alphabet_size ** kmer_length
    kmer_row = np.zeros(n_kmers, np.int32)
    for k in range(n_kmers):
        d = k
        while row_of[d] < 0:
            d = parent[d]
        kmer_row[k] = row_of[d]
    n = grid_length - 1
    dx = (grid_stop - grid_start) / float(n)   # linspace, impl/hdp_math_utils.c:497-510
    grid = np.array([grid_start + j * dx for j in range(n)] + [grid_stop])
    return dict(alphabet=alphabet, alphabet_size=alphabet_size, kmer_length=kmer_length, grid=grid,
                y=np.ascontiguousarray(np.stack(rows)), slope=np.ascontiguousarray(np.stack(slopes)),
                kmer_row=kmer_row)


class HdpModel:
    """3-state HDP signal model (getHdpStateMachine3, stateMachine.c:1738) over a parsed .nhdp"""

    def __init__(self, nhdp, transitions=None):
        assert nhdp["kmer_length"] == 6
        self.nhdp = nhdp
        self.c = OrcModel()
        lib().orc_defaults_hdp(C.byref(self.c))
        if transitions is not None:
            for i, v in enumerate(transitions):
                self.c.t[i] = v
        self.c.hdpRow = nhdp["kmer_row"].ctypes.data
        self.c.hdpGrid = nhdp["grid"].ctypes.data
        self.c.hdpY = nhdp["y"].ctypes.data
        self.c.hdpSlope = nhdp["slope"].ctypes.data
        self.c.gridLength = nhdp["grid"].size
        self.c.alphabetSize = nhdp["alphabet_size"]
        self.c.alphabet = nhdp["alphabet"].encode()

    @property
    def transitions(self):
        return np.array([self.c.t[i] for i in range(9)])

    def density(self, kmer, x):
        return lib().orc_hdp_density(C.byref(self.c), kmer.encode(), x)

    def kmer_id(self, kmer):
        return lib().orc_hdp_kmer_id(C.byref(self.c), kmer.encode())


class Sm5Model:
    def __init__(self):
        self.match = np.zeros(16)
        self.gx = np.zeros(4)
        self.gy = np.zeros(4)
        self.c = OrcModel()
        lib().orc_defaults_sm5(C.byref(self.c), _ptr(self.match), _ptr(self.gx), _ptr(self.gy))


def _collect(res):
    r = res.contents
    n = r.n
    tri = np.ctypeslib.as_array(r.triples, shape=(n, 3)).copy() if n else np.zeros((0, 3), np.int64)
    lp = np.ctypeslib.as_array(r.logp, shape=(n,)).copy() if n else np.zeros(0)
    nt = r.nTotals
    txay = np.ctypeslib.as_array(r.totalsXay, shape=(nt,)).copy() if nt else np.zeros(0, np.int64)
    tot = np.ctypeslib.as_array(r.totals, shape=(nt,)).copy() if nt else np.zeros(0)
    out = dict(triples=tri, logp=lp, totals_xay=txay, totals=tot, cells=int(r.cells))
    lib().orc_result_free(res)
    return out


def _xy(model, x, y):
    xb = x.encode() if isinstance(x, str) else bytes(x)
    if model.c.kind != SM5:
        ya = np.ascontiguousarray(y, dtype=np.float64).reshape(-1)
        return xb, ya, ya.ctypes.data_as(C.c_void_p), ya.size // 3
    yb = y.encode() if isinstance(y, str) else bytes(y)
    buf = C.create_string_buffer(yb)
    return xb, buf, C.cast(buf, C.c_void_p), len(yb)


def aligned_pairs_using_anchors(model, x, lX, y, anchors, params, ragged_left=False,
                                ragged_right=False, expectations=None):
    """getAlignedPairsUsingAnchors / getExpectationsUsingAnchors on the CPU oracle."""
    xb, keep, yptr, lY = _xy(model, x, y)
    a = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 2)
    res = lib().orc_result_new()
    eptr = C.byref(expectations) if expectations is not None else None
    rc = lib().orc_aligned_pairs_using_anchors(C.byref(model.c), xb, lX, yptr, lY, _ptr(a),
                                               a.shape[0], C.byref(params), int(ragged_left),
                                               int(ragged_right), eptr, res)
    out = _collect(res)
    if rc != 0:
        raise RuntimeError("oracle failed rc=%d" % rc)
    return out


def expectations_h_using_anchors(model, reads, params, threshold, ragged_left=False, ragged_right=False):
    """getExpectationsUsingAnchors for the HDP signal machine over reads = [(x, lX, events, anchors), ...]:
    dict(transitions[9], likelihood, assign[n][3] = (from, X index, Y index), logp[n])."""
    h = lib().orc_expectations_h_new(threshold)
    try:
        for x, lX, y, anchors in reads:
            xb, keep, yptr, lY = _xy(model, x, y)
            a = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 2)
            rc = lib().orc_expectations_h_using_anchors(C.byref(model.c), xb, lX, yptr, lY, _ptr(a), a.shape[0],
                                                        C.byref(params), int(ragged_left), int(ragged_right), h)
            if rc != 0:
                raise RuntimeError("oracle failed rc=%d" % rc)
        c = h.contents
        n = c.n
        return dict(transitions=np.array(c.transitions), likelihood=c.likelihood,
                    assign=np.ctypeslib.as_array(c.assign, shape=(n, 3)).copy() if n else np.zeros((0, 3), np.int64),
                    logp=np.ctypeslib.as_array(c.logp, shape=(n,)).copy() if n else np.zeros(0))
    finally:
        lib().orc_expectations_h_free(h)


def expectations_v_using_anchors(model, x, lX, y, anchors, params, hmm, ragged_left=False, ragged_right=False):
    """getExpectationsUsingAnchors for the vanilla signal machine: adds this alignment to hmm."""
    xb, keep, yptr, lY = _xy(model, x, y)
    a = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 2)
    rc = lib().orc_expectations_v_using_anchors(C.byref(model.c), xb, lX, yptr, lY, _ptr(a), a.shape[0],
                                                C.byref(params), int(ragged_left), int(ragged_right),
                                                C.byref(hmm))
    if rc != 0:
        raise RuntimeError("oracle failed rc=%d" % rc)
    return hmm


def expectations5_using_anchors(model, x, lX, y, anchors, params, hmm, ragged_left=False, ragged_right=False):
    """getExpectationsUsingAnchors for the 5-state symbol machine: adds this alignment to hmm."""
    xb, keep, yptr, lY = _xy(model, x, y)
    a = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 2)
    rc = lib().orc_expectations5_using_anchors(C.byref(model.c), xb, lX, yptr, lY, _ptr(a), a.shape[0],
                                               C.byref(params), int(ragged_left), int(ragged_right),
                                               C.byref(hmm))
    if rc != 0:
        raise RuntimeError("oracle failed rc=%d" % rc)
    return hmm


def aligned_pairs_without_banding(model, x, lX, y, params, ragged_left=False, ragged_right=False):
    xb, keep, yptr, lY = _xy(model, x, y)
    res = lib().orc_result_new()
    rc = lib().orc_aligned_pairs_without_banding(C.byref(model.c), xb, lX, yptr, lY,
                                                 C.byref(params), int(ragged_left),
                                                 int(ragged_right), res)
    out = _collect(res)
    if rc != 0:
        raise RuntimeError("oracle failed rc=%d" % rc)
    return out


def band(anchors, lX, lY, expansion):
    a = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 2)
    L = np.zeros(lX + lY + 1, np.int64)
    R = np.zeros(lX + lY + 1, np.int64)
    rc = lib().orc_band(_ptr(a), a.shape[0], lX, lY, expansion, _ptr(L), _ptr(R))
    if rc != 0:
        raise ValueError("invalid diagonal")
    return L, R


def banded_dump(model, x, lX, y, anchors, params, ragged_left=False, ragged_right=False):
    """One getPosteriorProbsWithBanding call, returning forward/backward cell dumps too."""
    xb, keep, yptr, lY = _xy(model, x, y)
    a = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 2)
    L, R = band(a, lX, lY, params.diagonalExpansion)
    widths = (R - L) // 2 + 1
    S = model.c.stateNumber
    F = np.full((int(widths.sum()), S), np.nan)
    B = np.full((int(widths.sum()), S), np.nan)
    res = lib().orc_result_new()
    rc = lib().orc_banded_dump(C.byref(model.c), xb, lX, yptr, lY, _ptr(a), a.shape[0],
                               C.byref(params), int(ragged_left), int(ragged_right), _ptr(F),
                               _ptr(B), res)
    out = _collect(res)
    if rc != 0:
        raise RuntimeError("oracle failed rc=%d" % rc)
    out.update(F=F, B=B, L=L, R=R, offsets=np.concatenate([[0], np.cumsum(widths)]))
    return out


def split_points(anchors, lX, lY, max_matrix, ragged_left, ragged_right):
    a = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 2)
    out = np.zeros((a.shape[0] + 2, 4), np.int64)
    n = lib().orc_split_points(_ptr(a), a.shape[0], lX, lY, max_matrix, int(ragged_left),
                               int(ragged_right), _ptr(out), out.shape[0])
    return out[:n]


def load_pore_model(path):
    """Parse a 3-line .model file (layout of emissions_signal_loadPoreModel, stateMachine.c:242):
    line 1 match table [1+4096*5], line 2 30 skip bins, line 3 extra-event table [1+4096*5]."""
    with open(path) as f:
        lines = f.read().split("\n")
    match = np.array(lines[0].split(), dtype=np.float64)
    skip = np.array(lines[1].split(), dtype=np.float64)
    gapy = np.array(lines[2].split(), dtype=np.float64)
    assert match.size == MODEL_LEN and gapy.size == MODEL_LEN and skip.size == 30
    return match, skip, gapy


def load_npread(path):
    """Parse a .npRead file (layout of nanopore_loadNanoporeReadFromFile, impl/nanopore.c:40)."""
    with open(path) as f:
        lines = f.read().split("\n")
    h = lines[0].split()
    out = dict(read_length=int(h[0]), n_template=int(h[1]), n_complement=int(h[2]),
               template_params=[float(v) for v in h[3:8]],
               complement_params=[float(v) for v in h[8:13]],
               read=lines[1].strip(),
               template_map=np.array(lines[2].split(), dtype=np.int64),
               template_events=np.array(lines[3].split(), dtype=np.float64),
               complement_map=np.array(lines[4].split(), dtype=np.int64),
               complement_events=np.array(lines[5].split(), dtype=np.float64))
    assert out["template_events"].size == 3 * out["n_template"]
    assert out["complement_events"].size == 3 * out["n_complement"]
    return out

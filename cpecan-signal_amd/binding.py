"""ctypes view of the C-ABI in include/cpecan_hip.h (libcpecan_hip.so), for tests and bench.py.

This file adds no computation: every call goes straight through the C-ABI.  It never imports the
oracle and has no CPU fallback -- without a GPU, Context() raises CpecanError(CPECAN_ENODEVICE).
The directory name contains a hyphen, so load this module by path:

    import importlib.util, os
    spec = importlib.util.spec_from_file_location("cpecan_binding", ".../cpecan-signal_amd/binding.py")
"""
import ctypes as C
import os
import sys
import weakref
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Two batches in flight use six streams (per context: its own, the input preparation's, the backward sweeps'); with the
# runtime's default of four hardware queues per process two of them would share a queue and a batch's forward and
# backward sweeps would take turns instead of overlapping.  (Read by the HIP runtime when it starts.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
LIB_PATH = os.environ.get("CPECAN_HIP_LIB") or os.path.join(_HERE, "libcpecan_hip.so")  # (the override serves tools/ablate_asm.sh, tools/ab_bench.sh)

OK, ENODEVICE, EINVAL, EHIP, EOVERFLOW, EBAND = 0, -1, -2, -3, -4, -5
MODE_POSTERIOR, MODE_EXPECTATIONS = 0, 1
KERNEL_AUTO, KERNEL_GENERAL, KERNEL_SYSTOLIC = 0, 1, 2
FLAG_DEBUG_DUMP = 1
FLAG_UNBANDED = 2
FLAG_SCAN_DECODE = 4
FLAG_SMALL_FOOTPRINT = 64
FLAG_EXPECTATIONS = 8
FLAG_WORKGROUP_KERNELS = 16
FLAG_GENERAL_KERNEL = 32
NUM_KMERS = 4096
MODEL_TABLE_LEN = 1 + NUM_KMERS * 5
EXPECTATION_LEN = 9 + NUM_KMERS + 1
EXPECTATION5_LEN = 25 + 5 * 16 + 1
EXPECTATIONV_LEN = 60 + 1
EXPECTATIONH_LEN = 9 + 1

# every symbol include/cpecan_hip.h declares
EXPORTS = [
    "cpecan_hip_device_count", "cpecan_hip_ctx_create", "cpecan_hip_ctx_destroy",
    "cpecan_hip_last_error", "cpecan_hip_version", "cpecan_hip_models_create",
    "cpecan_hip_models_clear", "cpecan_hip_models_create_scaled", "cpecan_hip_models_download", "cpecan_band_construct", "cpecan_split_points",
    "cpecan_hip_batch_create", "cpecan_hip_batch_run", "cpecan_hip_batch_run_after", "cpecan_hip_batch_shader_clock_mhz", "cpecan_hip_batch_sync",
    "cpecan_hip_batch_elapsed_ms", "cpecan_hip_batch_counts", "cpecan_hip_batch_fetch_pairs",
    "cpecan_hip_batch_fetch_totals", "cpecan_hip_batch_expectations_device_ptr",
    "cpecan_hip_batch_fetch_expectations", "cpecan_hip_batch_debug_cells",
    "cpecan_hip_batch_destroy", "cpecan_hip_ctx_stream", "cpecan_hip_selftest_division", "cpecan_hip_batch_info", "cpecan_hip_batch_stage_ms",
    "cpecan_hip_batch_systolic_rows", "cpecan_hip_batch_kernel_family", "cpecan_hip_batch_assembly_sweeps", "cpecan_hip_trim_cache", "cpecan_hip_models_set_transitions",
    "cpecan_hip_models5_create", "cpecan_hip_batch_create_dna",
    "cpecan_hip_modelsv_create", "cpecan_hip_batch_create_vanilla", "cpecan_hip_models4_create", "cpecan_hip_batch_create_sm4",
    "cpecan_hip_modelsh_create", "cpecan_hip_batch_create_hdp",
]


class CpecanError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cpecan_hip error %d: %s" % (code, msg))
        self.code = code


class Sm3ModelDesc(C.Structure):
    _fields_ = [("transitions", C.c_double * 9), ("match_probs", C.c_void_p),
                ("gap_x_probs", C.c_void_p), ("gap_y_probs", C.c_void_p)]


class Sm5ModelDesc(C.Structure):
    _fields_ = [("transitions", C.c_double * 17), ("match_probs", C.c_double * 16),
                ("gap_x_probs", C.c_double * 4), ("gap_y_probs", C.c_double * 4)]


class Sm4ModelDesc(C.Structure):
    _fields_ = [("transitions", C.c_double * 11), ("match_probs", C.c_void_p), ("gap_x_probs", C.c_void_p),
                ("gap_y_probs", C.c_void_p)]


class VanillaModelDesc(C.Structure):
    _fields_ = [("m_to_y_not_x", C.c_double), ("e_to_e", C.c_double), ("end_match_prob", C.c_double),
                ("end_from_x_prob", C.c_double), ("end_from_y_prob", C.c_double),
                ("match_probs", C.c_void_p), ("skip_probs", C.c_void_p), ("gap_y_probs", C.c_void_p)]


class HdpModelDesc(C.Structure):
    _fields_ = [("transitions", C.c_double * 9), ("alphabet", C.c_char_p), ("alphabet_size", C.c_int32),
                ("grid_length", C.c_int32), ("grid", C.c_void_p), ("n_rows", C.c_int64),
                ("posterior_predictive", C.c_void_p), ("spline_slopes", C.c_void_p), ("kmer_row", C.c_void_p)]


class Item(C.Structure):
    _fields_ = [("x_offset", C.c_int64), ("lX", C.c_int64), ("y_offset", C.c_int64),
                ("lY", C.c_int64), ("anchor_offset", C.c_int64), ("n_anchors", C.c_int64),
                ("model_id", C.c_int32), ("ragged_left", C.c_int32), ("ragged_right", C.c_int32),
                ("reserved", C.c_int32)]


class BandParams(C.Structure):
    _fields_ = [("threshold", C.c_double), ("minDiagsBetweenTraceBack", C.c_int64),
                ("traceBackDiagonals", C.c_int64), ("diagonalExpansion", C.c_int64)]


ITEM_DTYPE = np.dtype([("x_offset", "<i8"), ("lX", "<i8"), ("y_offset", "<i8"), ("lY", "<i8"),
                       ("anchor_offset", "<i8"), ("n_anchors", "<i8"), ("model_id", "<i4"),
                       ("ragged_left", "<i4"), ("ragged_right", "<i4"), ("reserved", "<i4")])
assert ITEM_DTYPE.itemsize == C.sizeof(Item)

NANOPORE_TRANSITIONS = (  # stateMachine3_setTransitionsToNanoporeDefaults, stateMachine.c:1278
    -0.23552123624314988, -0.21880828092192281, -0.013406326748077823, -1.6269694202638481,
    -4.3187242127300092, -1.6269694202638481, -4.3187242127239411, float("-inf"), float("-inf"))

_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s", "libcpecan_hip.so"])
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise CpecanError(ENODEVICE, "libcpecan_hip.so is not built (run __graft_entry__.build())")
        if "torch" not in sys.modules and not os.environ.get("CPECAN_NO_TORCH_PRELOAD"):
            # PyTorch-ROCm ships its own HIP/HSA runtime.  A process that loads this library first and torch
            # later ends up with two HSA runtimes and torch sees no GPU (measured on the MI355X box); loaded
            # after torch, this library binds to the runtime torch brought.  Callers that never use torch.cuda
            # (plain C programs, CPECAN_NO_TORCH_PRELOAD=1) run on the system runtime.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        L.cpecan_hip_last_error.restype = C.c_char_p
        L.cpecan_hip_version.restype = C.c_char_p
        L.cpecan_split_points.restype = C.c_int64
        L.cpecan_split_points.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                          C.c_int, C.c_int, C.c_void_p, C.c_int64]
        L.cpecan_band_construct.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                            C.c_void_p, C.c_void_p]
        L.cpecan_hip_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.cpecan_hip_ctx_destroy.argtypes = [C.c_void_p]
        L.cpecan_hip_ctx_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.cpecan_hip_models_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.cpecan_hip_models_clear.argtypes = [C.c_void_p]
        L.cpecan_hip_models_create_scaled.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.cpecan_hip_models_download.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.cpecan_hip_models_set_transitions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cpecan_hip_selftest_division.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.POINTER(C.c_int64)]
        L.cpecan_hip_batch_create.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
            C.c_void_p, C.c_int64, C.POINTER(BandParams), C.c_int32, C.c_int32, C.c_int32,
            C.POINTER(C.c_void_p)]
        L.cpecan_hip_models5_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.cpecan_hip_modelsv_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.cpecan_hip_modelsh_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.cpecan_hip_batch_create_hdp.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
            C.c_void_p, C.c_int64, C.POINTER(BandParams), C.c_int32, C.POINTER(C.c_void_p)]
        L.cpecan_hip_models4_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        L.cpecan_hip_batch_create_sm4.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
            C.c_void_p, C.c_int64, C.POINTER(BandParams), C.c_int32, C.POINTER(C.c_void_p)]
        L.cpecan_hip_batch_create_vanilla.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
            C.c_void_p, C.c_int64, C.POINTER(BandParams), C.c_int32, C.POINTER(C.c_void_p)]
        L.cpecan_hip_batch_create_dna.argtypes = [
            C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
            C.c_void_p, C.c_int64, C.POINTER(BandParams), C.c_int32, C.POINTER(C.c_void_p)]
        L.cpecan_hip_batch_run_after.argtypes = [C.c_void_p, C.c_void_p]
        L.cpecan_hip_batch_shader_clock_mhz.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        for name in ("run", "sync", "destroy"):
            getattr(L, "cpecan_hip_batch_" + name).argtypes = [C.c_void_p]
        L.cpecan_hip_batch_elapsed_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.cpecan_hip_batch_systolic_rows.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        L.cpecan_hip_batch_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.cpecan_hip_batch_stage_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.cpecan_hip_batch_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cpecan_hip_batch_fetch_pairs.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
        L.cpecan_hip_batch_fetch_totals.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
        L.cpecan_hip_batch_expectations_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p),
                                                               C.POINTER(C.c_int64)]
        L.cpecan_hip_batch_fetch_expectations.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        L.cpecan_hip_batch_debug_cells.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != OK:
        raise CpecanError(rc, lib().cpecan_hip_last_error().decode())


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    n = C.c_int(0)
    rc = lib().cpecan_hip_device_count(C.byref(n))
    return n.value if rc == OK else 0


def trim_cache():
    """gives the device and pinned host memory the library keeps for reuse back to the runtime (between phases, when
    something else in the process or on the card needs it)"""
    _check(lib().cpecan_hip_trim_cache())


def band_construct(anchors, lX, lY, expansion):
    a = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 2)
    L = np.zeros(lX + lY + 1, np.int32)
    R = np.zeros(lX + lY + 1, np.int32)
    _check(lib().cpecan_band_construct(_ptr(a), a.shape[0], lX, lY, expansion, _ptr(L), _ptr(R)))
    return L, R


def split_points(anchors, lX, lY, max_matrix, ragged_left, ragged_right):
    a = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 2)
    out = np.zeros((a.shape[0] + 2, 4), np.int64)
    n = lib().cpecan_split_points(_ptr(a), a.shape[0], lX, lY, max_matrix, int(ragged_left),
                                  int(ragged_right), _ptr(out), out.shape[0])
    return out[:n]


class Context:
    def __init__(self, device=0):
        h = C.c_void_p()
        _check(lib().cpecan_hip_ctx_create(device, C.byref(h)))
        self.h = h
        self.device = device
        self._batches = weakref.WeakSet()  # closed before the context is (a batch must not outlive its streams)

    def close(self):
        if self.h:
            for b in list(self._batches):
                b.close()
            lib().cpecan_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stream(self):
        s = C.c_void_p()
        _check(lib().cpecan_hip_ctx_stream(self.h, C.byref(s)))
        return s.value

    def models_set_transitions(self, transitions, gap_x=None):
        """the M-step's update of every strawMan model of the context, in place on the device"""
        t = np.ascontiguousarray(transitions, dtype=np.float64)
        assert t.size == 9
        g = None if gap_x is None else np.ascontiguousarray(gap_x, dtype=np.float64)
        assert g is None or g.size == NUM_KMERS
        _check(lib().cpecan_hip_models_set_transitions(self.h, _ptr(t), _ptr(g) if g is not None else None))

    def models_create(self, models, threads=0):
        """models: list of (transitions[9], match[20481], gap_x[4096], gap_y[20481]) -> ids"""
        n = len(models)
        descs = (Sm3ModelDesc * n)()
        keep = []
        for i, (t, match, gx, gy) in enumerate(models):
            match = np.ascontiguousarray(match, dtype=np.float64)
            gx = np.ascontiguousarray(gx, dtype=np.float64)
            gy = np.ascontiguousarray(gy, dtype=np.float64)
            assert match.size == MODEL_TABLE_LEN and gy.size == MODEL_TABLE_LEN and gx.size == NUM_KMERS
            keep += [match, gx, gy]
            for j in range(9):
                descs[i].transitions[j] = t[j]
            descs[i].match_probs = match.ctypes.data
            descs[i].gap_x_probs = gx.ctypes.data
            descs[i].gap_y_probs = gy.ctypes.data
        ids = np.zeros(n, np.int32)
        _check(lib().cpecan_hip_models_create(self.h, C.cast(descs, C.c_void_p), n, threads, _ptr(ids)))
        return ids

    def models_create_scaled(self, base, scalings, threads=0):
        """base: (transitions[9], match[20481], gap_x[4096], gap_y[20481]) of the unscaled pore model; scalings:
        [n, 5] (scale, shift, var, scale_sd, var_sd) per read -> ids.  emissions_signal_scaleModel per read, the
        rows assembled on the device."""
        t, match, gx, gy = base
        match = np.ascontiguousarray(match, dtype=np.float64)
        gx = np.ascontiguousarray(gx, dtype=np.float64)
        gy = np.ascontiguousarray(gy, dtype=np.float64)
        assert match.size == MODEL_TABLE_LEN and gy.size == MODEL_TABLE_LEN and gx.size == NUM_KMERS
        sc = np.ascontiguousarray(scalings, dtype=np.float64).reshape(-1, 5)
        desc = Sm3ModelDesc()
        for j in range(9):
            desc.transitions[j] = t[j]
        desc.match_probs, desc.gap_x_probs, desc.gap_y_probs = match.ctypes.data, gx.ctypes.data, gy.ctypes.data
        ids = np.zeros(len(sc), np.int32)
        _check(lib().cpecan_hip_models_create_scaled(self.h, C.byref(desc), _ptr(sc), len(sc), threads, _ptr(ids)))
        return ids

    def models_download(self, model_id):
        """the derived device table of one strawMan model (test aid)"""
        n = C.c_int64(0)
        _check(lib().cpecan_hip_models_download(self.h, 0, None, 0, C.byref(n)))
        out = np.zeros(n.value)
        _check(lib().cpecan_hip_models_download(self.h, int(model_id), _ptr(out), out.size, C.byref(n)))
        return out

    def models5_create(self, models):
        """models: list of (transitions[17], match[16], gap_x[4], gap_y[4]) -> ids (5-state symbol machine)"""
        n = len(models)
        descs = (Sm5ModelDesc * n)()
        for i, (t, match, gx, gy) in enumerate(models):
            for j in range(17):
                descs[i].transitions[j] = t[j]
            for j in range(16):
                descs[i].match_probs[j] = match[j]
            for j in range(4):
                descs[i].gap_x_probs[j] = gx[j]
                descs[i].gap_y_probs[j] = gy[j]
        ids = np.zeros(n, np.int32)
        _check(lib().cpecan_hip_models5_create(self.h, C.cast(descs, C.c_void_p), n, _ptr(ids)))
        return ids

    def modelsv_create(self, models, threads=0):
        """models: list of (scalars[5] = m_to_y_not_x, e_to_e, end_match, end_from_x, end_from_y;
        match[20481], skip[60], gap_y[20481]) -> ids (vanilla signal machine)"""
        n = len(models)
        descs = (VanillaModelDesc * n)()
        keep = []
        for i, (sc, match, skip, gy) in enumerate(models):
            match = np.ascontiguousarray(match, dtype=np.float64)
            skip = np.ascontiguousarray(skip, dtype=np.float64)
            gy = np.ascontiguousarray(gy, dtype=np.float64)
            assert match.size == MODEL_TABLE_LEN and gy.size == MODEL_TABLE_LEN and skip.size == 60
            keep += [match, skip, gy]
            (descs[i].m_to_y_not_x, descs[i].e_to_e, descs[i].end_match_prob, descs[i].end_from_x_prob,
             descs[i].end_from_y_prob) = [float(v) for v in sc]
            descs[i].match_probs = match.ctypes.data
            descs[i].skip_probs = skip.ctypes.data
            descs[i].gap_y_probs = gy.ctypes.data
        ids = np.zeros(n, np.int32)
        _check(lib().cpecan_hip_modelsv_create(self.h, C.cast(descs, C.c_void_p), n, threads, _ptr(ids)))
        return ids

    def models4_create(self, models):
        """models: list of (transitions[11] in the member order of _StateMachine4, match[20481], gap_x[4096],
        gap_y[20481]) -> ids (4-state signal machine, getStateMachine4)"""
        n = len(models)
        descs = (Sm4ModelDesc * n)()
        keep = []
        for i, (t, match, gx, gy) in enumerate(models):
            match = np.ascontiguousarray(match, dtype=np.float64)
            gx = np.ascontiguousarray(gx, dtype=np.float64)
            gy = np.ascontiguousarray(gy, dtype=np.float64)
            assert match.size == MODEL_TABLE_LEN and gy.size == MODEL_TABLE_LEN and gx.size == NUM_KMERS and len(t) == 11
            keep += [match, gx, gy]
            for k in range(11):
                descs[i].transitions[k] = float(t[k])
            descs[i].match_probs = match.ctypes.data
            descs[i].gap_x_probs = gx.ctypes.data
            descs[i].gap_y_probs = gy.ctypes.data
        ids = np.zeros(n, np.int32)
        _check(lib().cpecan_hip_models4_create(self.h, C.cast(descs, C.c_void_p), n, _ptr(ids)))
        return ids

    def modelsh_create(self, models):
        """models: list of (transitions[9], alphabet str, grid[G], y[rows, G], slope[rows, G],
        kmer_row[alphabet_size ** 6] int32) -> ids (HDP signal machine)"""
        n = len(models)
        descs = (HdpModelDesc * n)()
        keep = []
        for i, (t, alphabet, grid, y, slope, kmer_row) in enumerate(models):
            grid = np.ascontiguousarray(grid, dtype=np.float64)
            y = np.ascontiguousarray(y, dtype=np.float64)
            slope = np.ascontiguousarray(slope, dtype=np.float64)
            kmer_row = np.ascontiguousarray(kmer_row, dtype=np.int32)
            ab = alphabet.encode()
            assert y.shape == slope.shape and y.shape[1] == grid.size and kmer_row.size == len(ab) ** 6
            keep += [grid, y, slope, kmer_row, ab]
            for j in range(9):
                descs[i].transitions[j] = t[j]
            descs[i].alphabet = ab
            descs[i].alphabet_size = len(ab)
            descs[i].grid_length = grid.size
            descs[i].grid = grid.ctypes.data
            descs[i].n_rows = y.shape[0]
            descs[i].posterior_predictive = y.ctypes.data
            descs[i].spline_slopes = slope.ctypes.data
            descs[i].kmer_row = kmer_row.ctypes.data
        ids = np.zeros(n, np.int32)
        _check(lib().cpecan_hip_modelsh_create(self.h, C.cast(descs, C.c_void_p), n, _ptr(ids)))
        return ids

    def models_clear(self):
        _check(lib().cpecan_hip_models_clear(self.h))

    def selftest_division(self, n, seed=1):
        bad = C.c_int64(-1)
        _check(lib().cpecan_hip_selftest_division(self.h, n, seed, C.byref(bad)))
        return bad.value


class Batch:
    """cpecan_batch: items is a numpy array of ITEM_DTYPE."""

    def __init__(self, ctx, items, x_chars, events, anchors, params, mode=MODE_POSTERIOR,
                 kernel=KERNEL_AUTO, flags=0, y_chars=None, vanilla=False, hdp=False, sm4=False):
        """events: double[n][3] for a signal batch (vanilla: with a modelsv_create model); y_chars
        (str/bytes) instead for a DNA batch."""
        self.ctx = ctx
        items = np.ascontiguousarray(items, dtype=ITEM_DTYPE)
        xb = np.frombuffer(x_chars.encode() if isinstance(x_chars, str) else bytes(x_chars), np.uint8)
        an = np.ascontiguousarray(anchors, dtype=np.int64).reshape(-1, 2)
        h = C.c_void_p()
        if y_chars is not None:
            yb = np.frombuffer(y_chars.encode() if isinstance(y_chars, str) else bytes(y_chars), np.uint8)
            _check(lib().cpecan_hip_batch_create_dna(ctx.h, _ptr(items), items.shape[0], _ptr(xb), xb.size,
                                                     _ptr(yb), yb.size, _ptr(an), an.shape[0],
                                                     C.byref(params), flags, C.byref(h)))
        elif hdp:
            ev = np.ascontiguousarray(events, dtype=np.float64).reshape(-1)
            _check(lib().cpecan_hip_batch_create_hdp(ctx.h, _ptr(items), items.shape[0], _ptr(xb), xb.size,
                                                     _ptr(ev), ev.size // 3, _ptr(an), an.shape[0],
                                                     C.byref(params), flags, C.byref(h)))
        elif sm4:
            ev = np.ascontiguousarray(events, dtype=np.float64).reshape(-1)
            _check(lib().cpecan_hip_batch_create_sm4(ctx.h, _ptr(items), items.shape[0], _ptr(xb), xb.size,
                                                     _ptr(ev), ev.size // 3, _ptr(an), an.shape[0],
                                                     C.byref(params), flags, C.byref(h)))
        elif vanilla:
            ev = np.ascontiguousarray(events, dtype=np.float64).reshape(-1)
            _check(lib().cpecan_hip_batch_create_vanilla(ctx.h, _ptr(items), items.shape[0], _ptr(xb), xb.size,
                                                         _ptr(ev), ev.size // 3, _ptr(an), an.shape[0],
                                                         C.byref(params), flags, C.byref(h)))
        else:
            ev = np.ascontiguousarray(events, dtype=np.float64).reshape(-1)
            _check(lib().cpecan_hip_batch_create(ctx.h, _ptr(items), items.shape[0], _ptr(xb), xb.size,
                                                 _ptr(ev), ev.size // 3, _ptr(an), an.shape[0],
                                                 C.byref(params), mode, kernel, flags, C.byref(h)))
        self.h = h
        ctx._batches.add(self)
        self.n = items.shape[0]
        self.dna = y_chars is not None
        self.vanilla = bool(vanilla) and not self.dna
        self.hdp = bool(hdp) and not self.dna

    def run(self, after=None):
        """after: a Batch of another context -- this batch's kernels start when that one's last run has finished"""
        if after is None:
            _check(lib().cpecan_hip_batch_run(self.h))
        else:
            _check(lib().cpecan_hip_batch_run_after(self.h, after.h))

    def sync(self):
        _check(lib().cpecan_hip_batch_sync(self.h))

    def shader_clock_mhz(self):
        """MHz during the last run's forward sweeps (wave kernels; 0.0 otherwise)"""
        v = C.c_double(0.0)
        _check(lib().cpecan_hip_batch_shader_clock_mhz(self.h, C.byref(v)))
        return v.value

    def elapsed_ms(self):
        a, k = C.c_float(), C.c_float()
        _check(lib().cpecan_hip_batch_elapsed_ms(self.h, C.byref(a), C.byref(k)))
        return a.value, k.value

    def stage_ms(self):
        f, k, n = C.c_float(), C.c_float(), C.c_int32()
        _check(lib().cpecan_hip_batch_stage_ms(self.h, C.byref(f), C.byref(k), C.byref(n)))
        return f.value, k.value, n.value

    def info(self):
        k, w, m = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().cpecan_hip_batch_info(self.h, C.byref(k), C.byref(w), C.byref(m)))
        out = dict(kernel={1: "general", 2: "systolic"}.get(k.value, str(k.value)), workgroups=w.value,
                   max_band_width=m.value)
        if self.dna:
            f = C.c_int32()
            _check(lib().cpecan_hip_batch_kernel_family(self.h, C.byref(f)))
            if f.value:
                out["family"] = "wave (5-state)"
        if k.value == KERNEL_SYSTOLIC:
            r = C.c_int32()
            _check(lib().cpecan_hip_batch_systolic_rows(self.h, C.byref(r)))
            out["waves_per_workgroup"] = r.value
            f = C.c_int32()
            _check(lib().cpecan_hip_batch_kernel_family(self.h, C.byref(f)))
            out["family"] = "wave" if f.value else "workgroup"
            if f.value:
                out["cells_per_lane"] = r.value
            a = C.c_int32()
            _check(lib().cpecan_hip_batch_assembly_sweeps(self.h, C.byref(a)))
            out["assembly_sweeps"] = a.value
        return out

    def counts(self):
        p = np.zeros(self.n, np.int64)
        t = np.zeros(self.n, np.int64)
        c = np.zeros(self.n, np.int64)
        _check(lib().cpecan_hip_batch_counts(self.h, _ptr(p), _ptr(t), _ptr(c)))
        return p, t, c

    def pairs(self, item, n):
        tri = np.zeros((max(int(n), 1), 3), np.int64)
        lp = np.zeros(max(int(n), 1), np.float64)
        _check(lib().cpecan_hip_batch_fetch_pairs(self.h, item, _ptr(tri), _ptr(lp), tri.shape[0]))
        return tri[:n], lp[:n]

    def totals(self, item, n):
        xay = np.zeros(max(int(n), 1), np.int64)
        tot = np.zeros(max(int(n), 1), np.float64)
        _check(lib().cpecan_hip_batch_fetch_totals(self.h, item, _ptr(xay), _ptr(tot), xay.size))
        return xay[:n], tot[:n]

    def expectations(self, model_id):
        out = np.zeros(EXPECTATION5_LEN if self.dna else EXPECTATIONV_LEN if self.vanilla
                       else EXPECTATIONH_LEN if self.hdp else EXPECTATION_LEN)
        _check(lib().cpecan_hip_batch_fetch_expectations(self.h, int(model_id), _ptr(out)))
        return out

    def expectations_device_ptr(self):
        p, n = C.c_void_p(), C.c_int64()
        _check(lib().cpecan_hip_batch_expectations_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def debug_cells(self, item, n_cells):
        F = np.zeros((n_cells, 3))
        B = np.zeros((n_cells, 3))
        _check(lib().cpecan_hip_batch_debug_cells(self.h, item, _ptr(F), _ptr(B), n_cells))
        return F, B

    def close(self):
        if self.h:
            lib().cpecan_hip_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

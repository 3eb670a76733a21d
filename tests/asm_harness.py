"""The assembly sweeps on the CPU (test infrastructure): a Python replica of what the C-ABI layer prepares for them
(model rows, track, band plan, start contexts, argument block: cpecan_hip.hip, cpecan_asm.hip), the kernels run on the
instruction emulator (gcn_emu.py), and their outputs are compared with the oracle's cell dumps."""
import math
import os
import struct
import subprocess
import sys

import numpy as np

import gcn_emu as emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM_DIR = os.path.join(ROOT, "cpecan-signal_amd", "csrc", "asm")
if ASM_DIR not in sys.path:
    sys.path.insert(0, ASM_DIR)
import gen_sweeps as G  # noqa: E402  (the generator is also the single source of the formats)

import synth  # noqa: E402

NEG_INF = float("-inf")
MODEL_HEADER, ROW, NKMERS = 16, 18, 4096
MODEL_STRIDE = MODEL_HEADER + (NKMERS + 1) * ROW
STATE_BYTES = G.STATE_BYTES
L = G.L
P = 64 * L


def generated_text(cache={}):
    if "text" not in cache:
        out = os.path.join("/tmp", "cpecan_sweeps_test_%d.s" % os.getpid())
        subprocess.check_call([sys.executable, os.path.join(ASM_DIR, "gen_sweeps.py"), out], stderr=subprocess.DEVNULL)
        cache["text"] = open(out).read()
        cache["parsed"] = emu.parse(cache["text"])
        os.unlink(out)
    return cache["text"], cache["parsed"]


def derive_model(transitions, match, gap_x, gap_y):
    """derive_rows() of cpecan_hip.hip: mu, sd, 1/sd, K = -0.918... - log(sd) (host libm) per Gaussian; gap-X value"""
    m = np.zeros(MODEL_STRIDE)
    m[:9] = transitions
    rows = m[MODEL_HEADER:].reshape(NKMERS + 1, ROW)
    a = np.asarray(match)[1:].reshape(NKMERS, 5)
    b = np.asarray(gap_y)[1:].reshape(NKMERS, 5)
    c = -0.91893853320467267
    for g, (mu, sd) in enumerate(((a[:, 0], a[:, 1]), (a[:, 2], a[:, 3]), (b[:, 0], b[:, 1]), (b[:, 2], b[:, 3]))):
        rows[:NKMERS, 4 * g] = mu
        rows[:NKMERS, 4 * g + 1] = sd
        with np.errstate(divide="ignore"):
            rows[:NKMERS, 4 * g + 2] = np.where(sd == 0.0, 0.0, 1.0 / sd)
        rows[:NKMERS, 4 * g + 3] = [NEG_INF if s == 0.0 else c - math.log(s) for s in sd]
    rows[:NKMERS, 16] = gap_x
    rows[NKMERS, [3, 7, 11, 15, 16]] = NEG_INF
    return m


def track_rows(model, kidx, lX):
    """cpecan_k_wv_track: per matrix column the 16 emission constants of its k-mer, the gap-X sums, the gap-X value"""
    rows = model[MODEL_HEADER:].reshape(NKMERS + 1, ROW)
    t = np.zeros((lX + 1, 20))
    k = np.concatenate([[NKMERS], kidx[:lX]]).astype(np.int64)
    t[:, :16] = rows[k, :16]
    gx = rows[k, 16]
    t[:, 16] = gx + model[3]
    t[:, 17] = gx + model[5]
    t[:, 18] = gx + model[7]
    t[:, 19] = gx
    return t


def band_table(anchors, lX, lY, expansion):
    import pyoracle as o
    Lb, Rb = o.band(anchors, lX, lY, expansion)
    k = np.arange(lX + lY + 1)
    return ((k + Lb) // 2).astype(np.int64), ((k + Rb) // 2).astype(np.int64), Lb, Rb


def build_plan(xmin, xmax, min_diags, tb_diags, expansion):
    """build_asm_plan() of cpecan_hip.hip: the windows and the control words of one alignment"""
    D = len(xmin) - 1
    nblocks = D // G.BLOCK + 2
    ctl = np.zeros((nblocks, 4), np.uint64)
    wins = []
    traced, cells, d0 = 0, 1, 0
    for k in range(1, D + 1):
        if xmin[k] != xmin[k - 1]:
            ctl[k >> 6, 0] |= np.uint64(1 << (k & 63))
        if xmax[k] != xmax[k - 1]:
            ctl[k >> 6, 1] |= np.uint64(1 << (k & 63))
        w = int(xmax[k] - xmin[k] + 1)
        cells += w
        at_end = k == D
        if not (at_end or (k >= traced + min_diags and w <= expansion * 2 + 1)):
            continue
        frm = k - (0 if at_end else tb_diags + 1)
        wins.append(dict(d0=d0, top=k, frm=frm, to=traced, atEnd=int(at_end), xminTop=int(xmin[k]), xmaxTop=int(xmax[k]),
                         cells=cells, xmin0=int(xmin[d0]), xmax0=int(xmax[d0]), tpost0=min(k, frm)))
        d0, traced = k, frm
    for wi, w in enumerate(wins):
        w["nWindows"] = len(wins)
        endw = bool(w["atEnd"])
        tpa, tpb, allfull = w["tpost0"], w["tpost0"], False
        if not endw:
            tpb = wins[wi + 1]["tpost0"]
            allfull = tpb < w["top"]
        for dj in range(w["d0"] + 1, w["top"] + 1):
            ra = (tpa - dj) % 10
            rb = 99 if endw else (tpb - dj) % 10
            here = ra if dj <= w["frm"] else rb
            if allfull or dj >= w["top"] - 1 or here == 0:  # (the assembly sweep back reads no gap states below a refresh)
                ctl[dj >> 6, 2] |= np.uint64(1 << (dj & 63))
    return wins, ctl


def pack_win(w):
    return struct.pack("<8iq3i3i", w["d0"], w["top"], w["frm"], w["to"], w["atEnd"], w["xminTop"], w["xmaxTop"], w["nWindows"],
                       w["cells"], w["xmin0"], w["xmax0"], w["tpost0"], 0, 0, 0)


def start_context(ragged_left):
    """cpecan_k_asm_ctx_init: the registers of a forward wave that has done diagonal 0"""
    c = np.zeros(G.CTX_BYTES, np.uint8)
    d = c.view(np.float64)
    for j in range(L):
        for q in range(G.NCONST // 2):
            lo = NEG_INF if 2 * q == 16 else 0.0
            hi = NEG_INF if 2 * q + 1 in (3, 7, 11, 15, 17) else 0.0
            base = (9 * j + q) * 1024 // 8
            d[base: base + 128: 2] = lo
            d[base + 1: base + 128: 2] = hi
    x = G.CTX_X // 8
    d[x: x + 2 * L * 1536 // 8] = NEG_INF
    d[x + 0] = NEG_INF if ragged_left else 0.0                 # m of (parity 0, layer 0), lane 0
    d[x + 64 + 0] = d[x + 64 + 1] = 0.0 if ragged_left else NEG_INF  # (x, y)
    s = c[G.CTX_S:].view(np.int32)
    s[:10] = [1, 0, 0, 0, 0, 0, 1 // L, 1 % L, 0, 0]
    return c


class Image:
    """Device memory for one batch of the synthetic generator, as the library lays it out."""

    def __init__(self, batch, bp, ragged, transitions):
        self.batch, self.bp, self.ragged = batch, bp, ragged
        mem = self.mem = emu.Memory()
        items = batch["items"]
        n = self.n = len(items)
        self.models = np.concatenate([derive_model(transitions, *m) for m in batch["models"]])
        self.bands, self.plans = [], []
        ctl_all, plan_off, track_all, track_base = [], [], [], []
        max_span, max_windows = 1, 0
        for it in items:
            an = batch["anchors"][it["anchor_offset"]: it["anchor_offset"] + it["n_anchors"]]
            xmin, xmax, Lb, Rb = band_table(an, it["lX"], it["lY"], bp.diagonalExpansion)
            wins, ctl = build_plan(xmin, xmax, bp.minDiagsBetweenTraceBack, bp.traceBackDiagonals, bp.diagonalExpansion)
            self.bands.append((xmin, xmax, Lb, Rb))
            self.plans.append((wins, ctl))
            plan_off.append(sum(len(c) for c in ctl_all))
            ctl_all.append(ctl)
            max_windows = max(max_windows, len(wins))
            for w in wins:
                max_span = max(max_span, w["top"] - w["to"] + 1)
            seq = batch["x_chars"][it["x_offset"]: it["x_offset"] + it["lX"] + 5]
            track_base.append(sum(len(t) for t in track_all))
            model = self.models[it["model"] * MODEL_STRIDE: (it["model"] + 1) * MODEL_STRIDE]
            track_all.append(track_rows(model, synth.kmer_indices(seq), it["lX"]))
        self.max_windows = max(max_windows, 1)
        self.ringD = 64
        while self.ringD < (2 * max_span + 8 if max_windows > 1 else max_span + 4):
            self.ringD *= 2
        self.ring_doubles = (self.ringD + 1) * G.ROW_DOUBLES
        # buffers
        dev_items = np.zeros((n, 16), np.int64)
        for i, it in enumerate(items):
            dev_items[i, 0], dev_items[i, 1], dev_items[i, 3] = it["lX"], it["lY"], it["y_offset"]
            dev_items[i, 14] = it["model"] | (int(ragged[0]) << 32)
            dev_items[i, 15] = int(ragged[1])
        self.a_items = mem.alloc(dev_items.nbytes)
        mem.put(self.a_items, dev_items)
        self.a_trackbase = mem.alloc(8 * n)
        mem.put(self.a_trackbase, np.array(track_base, np.int64))
        planwin = bytearray(n * self.max_windows * G.PLANWIN_BYTES)
        for i, (wins, _) in enumerate(self.plans):
            for wi, w in enumerate(wins):
                o = (i * self.max_windows + wi) * G.PLANWIN_BYTES
                planwin[o: o + G.PLANWIN_BYTES] = pack_win(w)
        self.a_planwin = mem.alloc(len(planwin))
        mem.put(self.a_planwin, np.frombuffer(bytes(planwin), np.uint8))
        ctl = np.concatenate(ctl_all)
        self.a_planctl = mem.alloc(ctl.nbytes)
        mem.put(self.a_planctl, ctl)
        self.a_planoff = mem.alloc(8 * n)
        mem.put(self.a_planoff, np.array(plan_off, np.int64))
        ev = np.ascontiguousarray(batch["events"], np.float64)
        self.a_events = mem.alloc(ev.nbytes + 64)
        mem.put(self.a_events, ev)
        self.a_models = mem.alloc(self.models.nbytes)
        mem.put(self.a_models, self.models)
        track = np.concatenate(track_all)
        self.a_track = mem.alloc(track.nbytes)
        mem.put(self.a_track, track)
        self.a_ring = mem.alloc(n * self.ring_doubles * 8)
        self.a_states = mem.alloc(n * STATE_BYTES)
        self.a_ctx = mem.alloc(n * 3 * G.CTX_BYTES)
        nw = self.ringD // 10 + 8
        self.scratch_bytes = (2 * self.ringD * 4 + nw * (32 + 7 * P * 8) + 4 * self.ringD * 8
                              + L * 4 * self.ringD * 16 + 63) // 64 * 64
        self.a_scratch = mem.alloc(n * self.scratch_bytes)
        coef = np.zeros(64)
        t = np.array([-0.009350833524763, 0.130659527668286, 0.498799810682272, 0.693203116424741,
                      -0.014532321752540, 0.139942324101744, 0.495635523139337, 0.692140569840976,
                      -0.004605031767994, 0.063427417320019, 0.695956496475118, 0.514272634594009,
                      -0.000458661602210, 0.009695946122598, 0.930734667215156, 0.168037164329057], np.float32)
        for l in range(64):
            nn = l >> 2
            piece = 0 if nn <= 2 else 1 if nn <= 5 else 2 if nn <= 9 else 3
            coef[l] = float(t[piece * 4 + (l & 3)])
        self.a_coef = mem.alloc(512)
        mem.put(self.a_coef, coef)
        # the mask table: per diagonal the lanes of the band per layer, first and last column (cpecan_k_asm_masks)
        tabs, diag_base = [], []
        for (xmin, xmax, _, _) in self.bands:
            diag_base.append(sum(len(t) for t in tabs))
            tabs.append(mask_table(xmin, xmax))
        for i in range(n):
            dev_items[i, 6] = diag_base[i]
        mem.put(self.a_items, dev_items)
        tab = np.concatenate(tabs + [np.zeros((2, 16), np.uint32)])
        self.a_masktab = mem.alloc(tab.nbytes)
        mem.put(self.a_masktab, tab)
        self.a_args = mem.alloc(G.ARGS_BYTES)
        self.begin()

    def begin(self):
        """what a run starts from: states cleared, start contexts, ring row 0 and the -inf row"""
        mem = self.mem
        mem.put(self.a_states, np.zeros(self.n * STATE_BYTES, np.uint8))
        for i in range(self.n):
            mem.put(self.a_ctx + (3 * i + 2) * G.CTX_BYTES, start_context(self.ragged[0]))
            ring = np.full(self.ring_doubles, np.nan)
            ring[self.ringD * G.ROW_DOUBLES:] = NEG_INF
            ring[0] = NEG_INF if self.ragged[0] else 0.0
            ring[1] = ring[G.OFF_PY // 8] = 0.0
            ring[G.OFF_FXY // 8] = ring[G.OFF_FXY // 8 + 1] = 0.0 if self.ragged[0] else NEG_INF
            mem.put(self.a_ring + i * self.ring_doubles * 8, ring)

    def args(self, window, log_thr_slack=0.0):
        a = struct.pack("<14q4i2qdqq", self.a_items, self.a_trackbase, self.a_planwin, self.a_planctl, self.a_planoff,
                        self.a_events, self.a_models, self.a_track, self.a_ring, self.ring_doubles, self.a_states,
                        self.a_ctx, G.CTX_BYTES, self.a_coef, self.n, window, self.ringD, self.max_windows,
                        self.a_scratch, self.scratch_bytes, log_thr_slack, MODEL_STRIDE, self.a_masktab)
        assert len(a) == G.ARGS_BYTES
        self.mem.put(self.a_args, np.frombuffer(a, np.uint8))
        return self.a_args

    def ring_row(self, item, d):
        base = self.a_ring + item * self.ring_doubles * 8 + (d & (self.ringD - 1)) * G.ROW_BYTES
        return self.mem.get(base, np.float64, G.ROW_DOUBLES)

    def refreshes(self, item, n):
        """the first n refresh records of the window just swept back and their terms: [(t, xmin, xmax, nxmin, nxmax,
        second, v[P], w[P])], terms by slot"""
        base = self.a_scratch + item * self.scratch_bytes + 2 * self.ringD * 4
        nw = self.ringD // 10 + 8
        out = []
        for i in range(n):
            r = self.mem.get(base + 32 * i, np.int32, 8)
            vw = self.mem.get(base + 32 * nw + i * 2 * P * 8, np.float64, 2 * P)
            out.append((int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5]), vw[:P].copy(), vw[P:].copy()))
        return out

    def state(self, item):
        raw = self.mem.get(self.a_states + item * STATE_BYTES, np.uint8, STATE_BYTES)
        i32 = raw.view(np.int32)
        out = dict(d=int(i32[0]), tracedBackTo=int(i32[1]), finished=int(i32[2]), win=[])
        for k in range(4):
            w = raw[16 + 40 * k: 56 + 40 * k]
            wi = w.view(np.int32)
            out["win"].append(dict(valid=int(wi[0]), top=int(wi[1]), frm=int(wi[2]), to=int(wi[3]), atEnd=int(wi[4]),
                                   nCand=int(wi[5]), nRefresh=int(wi[6]), est=float(w[32:40].view(np.float64)[0])))
        out["cells"] = int(raw[G.ST_CELLS: G.ST_CELLS + 8].view(np.int64)[0])
        return out


def mask_table(xmin, xmax):
    """cpecan_k_asm_masks: per diagonal the u64 lanes of the band for each layer, the band's first and last column, and
    the lanes a ring row is stored / loaded under: the band and the slots next to it on either side, widened to groups of
    16 lanes (every other slot is parked in the forward sweep: -inf)"""
    tab = np.zeros((len(xmin), 16), np.uint32)
    for d in range(len(xmin)):
        m, g = [0] * L, [0] * L
        lo, hi = int(xmin[d]), int(xmax[d])
        for x in range(lo - 1, hi + 2):         # (column -1: the last slot, parked while the band starts at column 0)
            lane, j = (x % P) // L, (x % P) % L
            g[j] |= ((1 << G.MASK_GROUP) - 1) << (lane & ~(G.MASK_GROUP - 1))   # whole 128-byte lines: no partial writes
            if lo <= x <= hi:
                m[j] |= 1 << lane
        for j in range(L):
            tab[d, 2 * j], tab[d, 2 * j + 1] = m[j] & 0xFFFFFFFF, m[j] >> 32
            tab[d, 8 + 2 * j], tab[d, 9 + 2 * j] = g[j] & 0xFFFFFFFF, g[j] >> 32
        tab[d, 6], tab[d, 7] = lo, hi
        h = 0   # the last layer's 8-byte emissions: 16 lanes to a line
        for q in range(0, 64, 16):
            if g[L - 1] & (0xFFFF << q):
                h |= 0xFFFF << q
        tab[d, 14], tab[d, 15] = h & 0xFFFFFFFF, h >> 32
    return tab


def slot_of(x):
    s = x % P
    return s // L, s % L  # lane, layer


def run_backward(img, window, item, log_thr_slack, watch=None):
    text, parsed = generated_text()
    name = "cpecan_k_asm_backward_l%d" % L
    a = img.args(window, log_thr_slack)
    instrs, labels = parsed
    w = emu.Wave(instrs, labels, img.mem, emu.kernel_lds_bytes(text, name), a, item, name)
    if watch:
        w.watch = watch
    w.run()
    return w


def run_forward(img, window, item=None):
    text, parsed = generated_text()
    name = "cpecan_k_asm_forward_l%d" % L
    a = img.args(window)
    if item is None:
        return emu.run_kernel(text, name, img.mem, a, img.n, parsed=parsed)
    instrs, labels = parsed
    w = emu.Wave(instrs, labels, img.mem, emu.kernel_lds_bytes(text, name), a, item, name)
    w.run()
    return w

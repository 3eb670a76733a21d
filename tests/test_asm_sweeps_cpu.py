"""The hand-scheduled assembly sweeps (csrc/asm/gen_sweeps.py) on the instruction emulator (gcn_emu.py), no GPU: every
forward cell the forward kernel leaves in the ring and every backward cell the backward kernel holds in its registers
equal the oracle's cell dump (banded_dump) bit for bit, window by window, with the host preparation replicated by
asm_harness.py.  The emulator also enforces what the hardware does not forgive and no test on the card would pin down:
registers read or overwritten while a load into them is in flight, two scalar loads in flight to one register, DPP and
wide-store hazards, accesses outside the buffers the C-ABI layer allocates."""
import math

import numpy as np
import pytest

import asm_harness as H
import pyoracle
import synth
from harness import band_params, cp, run_oracle_item

G = H.G
BACKWARD = "cpecan_k_asm_backward_l3"


def _same(a, b):
    return a == b or (np.isnan(a) and np.isnan(b))


def _check_forward_rows(img, item, w, F, off):
    xmin, xmax, _, _ = img.bands[item]
    _, ctl = img.plans[item]
    bad = []
    for d in range(w["d0"] + 1, w["top"] + 1):
        row = img.ring_row(item, d)
        full = (int(ctl[d >> 6, 2]) >> (d & 63)) & 1
        for x in range(xmin[d], xmax[d] + 1):
            lane, j = H.slot_of(x)
            base = j * (G.LAYER_BYTES // 8)
            fxy = base + G.OFF_FXY // 8
            got = (row[base + 2 * lane], row[fxy + 2 * lane], row[fxy + 2 * lane + 1])
            want = F[off[d] + x - xmin[d]]
            ok = _same(got[0], want[0]) and (not full or (_same(got[1], want[1]) and _same(got[2], want[2])))
            if not ok:
                bad.append((d, x, got, tuple(want)))
    return bad


def _backward_hooks(img, item, cur, B, off, bad):
    xmin, xmax, _, _ = img.bands[item]

    def make(kk):
        def hook(w):
            t = int(w.s[8])
            win = cur[0]
            if t > win["frm"] or t <= win["to"]:
                return
            for x in range(xmin[t], xmax[t] + 1):
                lane, j = H.slot_of(x)

                def rd(r):
                    return np.array([int(w.v[r][lane]) | (int(w.v[r + 1][lane]) << 32)], np.uint64).view(np.float64)[0]

                got = (rd(G.B_M0 + 2 * G.L * kk + 2 * j), rd(G.B_BX0 + 2 * j), rd(G.B_BX0 + 2 * G.L + 2 * j))
                want = B[off[t] + x - xmin[t]]
                if not all(_same(g, r) for g, r in zip(got, want)):
                    bad.append((t, x, got, tuple(want)))
        return hook

    return {".L_%s_tail%d" % (BACKWARD, kk): make(kk) for kk in range(3)}


@pytest.mark.parametrize("lX,lY,ragged,seed,every", [(257, 330, (1, 0), 21, 50), (90, 420, (0, 1), 5, 50),
                                                     # no anchors: the whole matrix, a k-mer enters on every diagonal
                                                     (100, 230, (1, 1), 9, 10 ** 6)])
def test_sweeps_on_the_emulator_equal_the_oracle(lX, lY, ragged, seed, every):
    batch = synth.make_batch(seed, 1, lX, lY, anchor_every=every)
    bp = band_params(0.01, 150, 40, 100)
    img = H.Image(batch, bp, ragged, cp.NANOPORE_TRANSITIONS)
    ref = run_oracle_item(batch, 0, bp, ragged, dump=True)
    F, B, off = ref["F"], ref["B"], ref["offsets"]
    wins, _ = img.plans[0]
    assert len(wins) >= 3  # (first, middle and last windows all happen)
    assert lX <= 192 or max(int(w["xmaxTop"]) for w in wins) >= 192  # (the slots wrap when the read is long enough)
    thr = math.log(0.01) - 1e-3
    ladd = pyoracle.lib().orc_logAdd
    totals = {int(a): float(b) for a, b in zip(ref["totals_xay"], ref["totals"])}
    cur, bad_b = [None], []
    watch = _backward_hooks(img, 0, cur, B, off, bad_b)
    cells = 0
    for wi, w in enumerate(wins):
        H.run_forward(img, wi, 0)
        bad_f = _check_forward_rows(img, 0, w, F, off)
        assert not bad_f, "forward cells, window %d: %r" % (wi, bad_f[:4])
        cur[0] = w
        H.run_backward(img, wi, 0, thr, watch)
        assert not bad_b, "backward cells, window %d: %r" % (wi, bad_b[:4])
        rec = img.state(0)["win"][wi & 3]
        assert rec["valid"] == 3 and rec["top"] == w["top"] and rec["frm"] == w["frm"] and rec["to"] == w["to"]
        n_refresh = len([t for t in range(w["tpost0"], w["to"], -10)])
        assert rec["nRefresh"] == n_refresh
        # the refreshes' terms, folded as the post kernel folds them (diagonalCalculationTotalProbability, :736-754:
        # the cells of t by column, then the cells of t + 1), are the oracle's totalProbability of that diagonal
        for t, lo, hi, nlo, nhi, second, v, wt in img.refreshes(0, n_refresh):
            tot = float("-inf")
            for x in range(lo, hi + 1):
                tot = ladd(tot, float(v[x % H.P]))
            if second:
                acc = float("-inf")
                for x in range(nlo, nhi + 1):
                    acc = ladd(acc, float(wt[x % H.P]))
                tot = ladd(tot, acc)
            assert t in totals and tot == totals[t], "totalProbability of diagonal %d: %r, oracle %r" % (t, tot, totals.get(t))
    st = img.state(0)
    assert st["finished"] == 1 and st["d"] == lX + lY
    xmin, xmax, _, _ = img.bands[0]
    assert st["cells"] == int((xmax - xmin + 1).sum())


def test_table_row_by_round_up_fma_is_ceil_of_twice_the_difference():
    """ladd_rows() of gen_sweeps.py: the low word of fma(d, 2, 2^52) rounded towards +inf is ceil(2 d) -- at, just above and
    just below every limit of the reference's cubic pieces (impl/pairwiseAligner.c:238-249), for tiny and for random d."""
    import ctypes
    import math
    import struct
    import gcn_emu
    lib = gcn_emu._libm
    rng = np.random.default_rng(7)
    ds = [0.0, 5e-324, 1e-300, 2.0 ** -53, 7.5, 1e6, 2.0 ** 30]
    for lim in [0.5 * i for i in range(1, 16)]:
        ds += [lim, math.nextafter(lim, 0.0), math.nextafter(lim, 10.0)]
    ds += list(rng.uniform(0.0, 8.0, 2000)) + list(np.exp(rng.uniform(-40.0, 3.0, 2000)))
    lib.fesetround(gcn_emu._FE_UPWARD)
    try:
        got = [struct.unpack("<Q", struct.pack("<d", lib.fma(ctypes.c_double(d), ctypes.c_double(2.0), ctypes.c_double(2.0 ** 52))))[0]
               & 0xFFFFFFFF for d in ds]
    finally:
        lib.fesetround(gcn_emu._FE_TONEAREST)
    for d, n in zip(ds, got):
        assert n == math.ceil(2.0 * d), (d, n)

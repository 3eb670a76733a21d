#!/usr/bin/env python3
"""Generates the hand-scheduled gfx950 assembly of the sweeps of the banded pair-HMM (strawMan signal machine,
posterior decode): cpecan_k_asm_forward_l3 and cpecan_k_asm_backward_l3.

Same algorithm, same memory formats and -- bit for bit -- the same arithmetic as the compiled wave kernels
(cpecan_kernel_wave.hip: one wave per alignment, L cells per lane, slot = matrix column mod 64 L, lane = slot / L,
layer = slot % L); what changes is who schedules the instructions.  The compiler's sweeps spend 390 + 365 vector
instructions and about 300 + 350 scalar / wait / branch instructions per anti-diagonal; written out by hand a
diagonal takes about 350 + 280 vector instructions and a fifth of the others, and both sweeps leave room in a SIMD's
register file for the post-processing wave.

Everything irregular stays outside: the host plans the traceback windows and the band's edge steps per diagonal
(cpecan_hip.hip: build_asm_plan), a forward wave carries its registers from one launch to the next through a context
block in HBM instead of re-deriving them, and what follows a sweep back (the totals' terms, the folds, the decode) is
the compiled post kernel's.

Reference arithmetic: cell_calculateForward / Backward impl/pairwiseAligner.c:365-389, stateMachine3_cellCalculate
impl/stateMachine.c:1305-1334, logAdd / lookup impl/pairwiseAligner.c:235-255, emissions_signal_logGaussPdf
impl/stateMachine.c:333-343.

usage: gen_sweeps.py OUT.s [OUT.h]   (OUT.h: the sizes and offsets the C++ side shares, cpecan_asm_gen.h)
"""
import os
import struct
import sys

from emit import Emitter, Neg, Pool, S, V

L = 3
P = 64 * L

# ---------------------------------------------------------------- memory formats shared with the C++ side (cpecan_asm.h)
# a ring row (the layout of cpecan_kernel_wave.hip, WV_ROW_*): per layer (Fm, pm) x 64 | (Fx, Fy) x 64, then the gap-Y
# emissions of the layers two by two: (py0, py1) x 64 | (py2, unused) x 64 -- every access is 16 bytes per lane
LAYER_BYTES = 2 * 64 * 16
OFF_FXY = 64 * 16
OFF_PY = L * LAYER_BYTES                 # + 1024 per pair of layers
PY2_X2 = os.environ.get("CPECAN_ASM_PY2", "x2") == "x2"   # the odd last layer's emissions as 8 bytes per lane (less traffic) or
                                                           # padded to 16 (fewer, faster instructions)
ROW_BYTES = L * LAYER_BYTES + 1024 + (512 if PY2_X2 else 1024)
ROW_DOUBLES = ROW_BYTES // 8
assert L == 3                            # (the pairing of the gap-Y emissions below is written for three layers)
TRACK_ROW_BYTES = 20 * 8                 # a track row: 16 emission constants, gap-X sums (open, extend, switch), gap-X
NCONST = 18                              # ... of which a slot keeps the first 18 doubles

# kernel arguments: struct AsmArgs
A_ITEMS, A_TRACKBASE, A_PLANWIN, A_PLANCTL, A_PLANOFF, A_EVENTS, A_MODELS, A_TRACK = 0, 8, 16, 24, 32, 40, 48, 56
A_RING, A_RINGDOUBLES, A_STATES, A_CTX, A_CTXBYTES, A_COEF, A_NITEMS, A_WINDOW = 64, 72, 80, 88, 96, 104, 112, 116
A_RINGD, A_MAXWIN, A_SCRATCH, A_SCRATCHBYTES, A_LOGTHR, A_MODELSTRIDE, ARGS_BYTES = 120, 124, 128, 136, 144, 152, 168
A_MASKTAB = 160                          # per diagonal of every item: the band's lanes per layer (6 dwords), its first and last
                                         # column, the lanes a ring row is stored / loaded under (6 dwords): the band and its
                                         # two neighbour slots, whose k-mers are parked in the forward sweep: -inf emissions
# DevItem
I_LX, I_LY, I_YOFF, I_DIAGBASE, I_MODEL, I_RAGGEDL, I_RAGGEDR, ITEM_BYTES = 0, 8, 24, 48, 112, 116, 120, 128
# AsmPlanWin: 16 dwords
W_D0, W_TOP, W_FROM, W_TO, W_ATEND, W_XMINTOP, W_XMAXTOP, W_NWIN, W_CELLS, W_XMIN0, W_XMAX0, W_TPOST0 = \
    0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12
PLANWIN_BYTES = 64
CTL_BYTES = 32                           # per 64 diagonals: u64 stepMin, stepMax, full, spare
# WvState / WvWindow (cpecan_sweep.h)
ST_D, ST_TB, ST_FIN, ST_WIN, ST_CELLS, ST_CLKS, ST_CLKR, STATE_BYTES = 0, 4, 8, 16, 192, 200, 208, 216   # (WvState: win[4])
WIN_BYTES = 40
# the forward sweep's context (registers of a wave between two launches)
CTX_C = 0                                # L x 36 dwords of constants, 1024 bytes per 4 dwords
CTX_X = L * 9 * 1024                     # per (parity, layer): m (512 bytes), (x, y) (1024 bytes)
CTX_S = CTX_X + 2 * L * 1536             # scalars: masks, band slots
CTX_BYTES = CTX_S + 256

# LDS of the forward kernel
LDS_COEF = 0
LDS_EV = 512                             # events by index mod 256, mirrored: (2 * 256 + L) x 16 bytes
EVN = 256
LDS_ROWS = LDS_EV + ((2 * EVN + L) * 16 + 63) // 64 * 64
ROWN = 64
LDS_F_BYTES = LDS_ROWS + ROWN * TRACK_ROW_BYTES
# ... and of the backward kernel
LDS_PX = 512
PXN = 64
LDS_RF = LDS_PX + PXN * 16               # the forward cells a refresh of totalProbability wants, fetched ahead by loads to LDS:
LDS_RF_XY_T = LDS_RF                     # (Fx, Fy) of the refresh's diagonal t, 1024 bytes per layer
LDS_B_BYTES = LDS_RF + L * 1024

MAX_WIDTH = 158                          # band widths the staging scheme holds
BLOCK = 64                               # diagonals per staging block

# timing studies (tools/ablate_asm.sh): wrong results by construction, never part of the product build
ABLATE = set(os.environ.get("CPECAN_ASM_ABLATE", "").split())

LOG2E_F32 = 0x3FB8AA3B
LN2_F32 = 0x3F317218


def dbits(x):
    return struct.unpack("<Q", struct.pack("<d", x))[0]


class Kernel(Emitter):
    """Emitter + the idioms both sweeps share."""

    def __init__(self, pool):
        super().__init__()
        self.pool = pool

    def ror64(self, dst, src):
        for h in (0, 1):
            self.valu("v_mov_b32_dpp", dst.sub(h), src.sub(h), dpp="wave_ror:1 row_mask:0xf bank_mask:0xf")

    def rol64(self, dst, src):
        for h in (0, 1):
            self.valu("v_mov_b32_dpp", dst.sub(h), src.sub(h), dpp="wave_rol:1 row_mask:0xf bank_mask:0xf")

    def add(self, dst, a, b):
        self.valu("v_add_f64", dst, a, b)

    def mul(self, dst, a, b):
        self.valu("v_mul_f64", dst, a, b)

    # ---- logAdd in three parts (impl/pairwiseAligner.c:235-255; ladd() in cpecan_kernel_wave.hip): hi / lo are the
    # operands as the reference's two branches order them, d = hi - lo, the cubic's four float-literal coefficients
    # come from the LDS table at offset 0 indexed by ceil(2 d) (the pieces' limits are multiples of 1/2).  d >= 7.5,
    # an infinite or a NaN d discard the cubic: the index is then anything at all -- an LDS read beyond the allocation
    # returns zero (measured: profiles/r03_ubench2_exec_lds_branch.txt) -- so it is not clamped.
    def ladd_rows(self, recs):
        """The table rows of a group of logAdds.  n = ceil(2 d) picks the row: 2 d + 2^52 rounded towards +inf IS
        2^52 + n (doubles are a unit apart there; the product is exact, the fma rounds once, upwards), so the sum's low
        word holds n.  Only these fmas run under that rounding mode: everything else is the reference's round-to-nearest.
        (A difference of 2^31 or more, or infinite, gives an offset beyond the table: the row read there is not used,
        d >= 7.5; NaN -- both operands -inf -- gives whatever, unused as well.)  Written add / ceil / add this was three
        instructions, and v_ceil_f64 is not a fast one: the pass went from 32.7 to 30.8 ms when a timing build merely
        left it out (profiles/r03_ablate_valu.txt)."""
        p = self.pool
        for r in recs:
            r[3], r[4] = p.take(4), p.take(4)
        if "OLDIDX" in ABLATE:                 # (timing: the three-instruction form, with or without its v_ceil_f64)
            for r in recs:
                t = r[3].sub(0, 2)
                self.add(t, r[2], r[2])
                if "NOCEIL" not in ABLATE:
                    self.valu("v_ceil_f64_e32", t, t)
                self.add(r[4].sub(0, 2), t, self.magic)
                self.ds_read(128, r[3], r[4].lo)
                self.ds_read(128, r[4], r[4].lo, 16)
            return
        if "NOSETREG" not in ABLATE:           # (timing: what the two mode switches cost)
            self.salu("s_setreg_imm32_b32", "hwreg(HW_REG_MODE, 2, 2)", 1) # fp64 rounding: towards +inf
        for r in recs:
            self.valu("v_fma_f64", r[4].sub(0, 2), r[2], "2.0", self.magic)
        if "NOSETREG" not in ABLATE:
            self.salu("s_setreg_imm32_b32", "hwreg(HW_REG_MODE, 2, 2)", 0) # ... to nearest even again
        for r in recs:
            a = r[4].lo
            self.valu("v_lshlrev_b32_e32", a, 5, a)                        # 32 bytes a row
            if "COEF0" in ABLATE:
                self.valu("v_mov_b32_e32", a, 0)
            if "NOLDS" in ABLATE:
                for q in range(4):
                    self.valu("v_mov_b32_e32", r[3].sub(q), r[2].sub(q % 2))
                    self.valu("v_mov_b32_e32", r[4].sub(q), r[2].sub(q % 2))
                continue
            self.ds_read(128, r[3], a)
            if "HALFLDS" in ABLATE:
                for q in range(4):
                    self.valu("v_mov_b32_e32", r[4].sub(q), r[2].sub(q % 2))
                continue
            self.ds_read(128, r[4], a, 16)

    def ladd_back_group(self, recs, dsts):
        """The second halves of a group of independent logAdds, stage by stage: a wave issues a vector instruction
        that depends on the one before it 8.8 cycles after it, one that does not 5.8 (profiles/r03_ubench_dep.txt), and
        a logAdd's second half is a chain of ten."""
        rs = [rec[3].sub(0, 2) for rec in recs]
        for rec, r in zip(recs, rs):
            self.mul(r, rec[3].sub(0, 2), rec[2])
        for rec, r in zip(recs, rs):
            self.add(r, r, rec[3].sub(2, 2))
        for rec, r in zip(recs, rs):
            self.mul(r, r, rec[2])
        for rec, r in zip(recs, rs):
            self.add(r, r, rec[4].sub(0, 2))
        for rec, r in zip(recs, rs):
            self.mul(r, r, rec[2])
        for rec, r in zip(recs, rs):
            self.add(r, r, rec[4].sub(2, 2))
        for rec, r in zip(recs, rs):
            self.add(r, r, rec[1])
        # (the compares' outcomes in vcc, one logAdd after the other; kept in vcc and two SGPR pairs so that the selects
        # could be interleaved too it measured the same, profiles/r03_same_box_ab_runs.txt)
        for rec, r, dst in zip(recs, rs, dsts):
            self.valu("v_cmp_gt_f64_e32", "vcc", "0x401e0000", rec[2])      # 7.5 > d
            self.valu("v_cndmask_b32_e32", dst.lo, rec[0].lo, r.lo, "vcc")
            self.valu("v_cndmask_b32_e32", dst.hi, rec[0].hi, r.hi, "vcc")
        for rec in recs:
            self.pool.give(*rec)

    def need_recs(self, recs):
        """the table rows of a group of logAdds whose second halves follow: one wait"""
        self.need(*[r[4] for r in recs], *[r[3] for r in recs])

    # ---- log N(x; mu, sd) = K - (a / 2) a, a = (x - mu) / sd as a Markstein-corrected multiply by RN(1 / sd) (lgauss());
    # (a / 2) a = RN(a a) / 2 exactly (but where a a is subnormal, which K absorbs whole), so the last two steps are one fma
    def gauss(self, dst, x, mu, sd, rsd, K, t0, t1):
        self.add(t0, x, Neg(mu))
        self.mul(t1, t0, rsd)
        self.valu("v_fma_f64", t0, Neg(t1), sd, t0)
        self.valu("v_fma_f64", t1, t0, rsd, t1)
        self.mul(t0, t1, t1)
        self.valu("v_fma_f64", dst, t0, "-0.5", K)   # K - (a / 2) a: halving is exact, so this is the same sum rounded once

    def gauss_group(self, items):
        """gauss() of independent arguments, stage by stage: items = (dst, x, mu, sd, rsd, K); dst serves as the first
        temporary, the second comes from the pool"""
        t1s = [self.pool.take(2) for _ in items]
        for (dst, x, mu, sd, rsd, K), t1 in zip(items, t1s):
            self.add(dst, x, Neg(mu))
        for (dst, x, mu, sd, rsd, K), t1 in zip(items, t1s):
            self.mul(t1, dst, rsd)
        for (dst, x, mu, sd, rsd, K), t1 in zip(items, t1s):
            self.valu("v_fma_f64", dst, Neg(t1), sd, dst)
        for (dst, x, mu, sd, rsd, K), t1 in zip(items, t1s):
            self.valu("v_fma_f64", t1, dst, rsd, t1)
        for (dst, x, mu, sd, rsd, K), t1 in zip(items, t1s):
            self.mul(dst, t1, t1)
        for (dst, x, mu, sd, rsd, K), t1 in zip(items, t1s):
            self.valu("v_fma_f64", dst, dst, "-0.5", K)
        self.pool.give(*t1s)

    def ladd_front_group(self, pairs, free=()):
        """the first halves of independent logAdds, stage by stage"""
        p = self.pool
        recs = [[p.take(2), p.take(2), p.take(2), None, None] for _ in pairs]
        for (x, y), r in zip(pairs, recs):
            self.valu("v_max_f64", r[0], x, y)
        for (x, y), r in zip(pairs, recs):
            self.valu("v_min_f64", r[1], x, y)
        p.give(*free)
        for r in recs:
            self.add(r[2], r[0], Neg(r[1]))
        return recs

    def s_mov64_lit(self, dst, value):
        self.salu("s_mov_b32", dst.lo, "0x%x" % (value & 0xFFFFFFFF))
        self.salu("s_mov_b32", dst.hi, "0x%x" % (value >> 32))

    def add64(self, dst, a, b_lo, b_hi=0):
        self.salu("s_add_u32", dst.lo, a.lo, b_lo)
        self.salu("s_addc_u32", dst.hi, a.hi, b_hi)

    def pc_of(self, dst, label):
        """dst = address of label (which lies AFTER this point)."""
        here = self.newlabel("pc")
        self.salu("s_getpc_b64", dst)
        self.label(here)
        self.salu("s_add_u32", dst.lo, dst.lo, "%s-%s" % (label, here))
        self.salu("s_addc_u32", dst.hi, dst.hi, 0)

    def butterfly(self, op, val, tmp, vLane4, tmp2):
        """all lanes <- op over the wave of val (32-bit), by ds_bpermute with lane ^ 1, 2, .. 32"""
        for off in (1, 2, 4, 8, 16, 32):
            self.valu("v_xor_b32_e32", tmp2, off * 4, vLane4)
            self.lines.append("\tds_bpermute_b32 %s, %s, %s" % (tmp, tmp2, val))
            self.n += 1
            self.lgkm.append({tmp.i})
            self.valu(op, val, val, tmp)


# ====================================================================================================== forward
def forward_kernel(name):
    """The forward sweep of one traceback window, one wave per alignment (forward_window() of cpecan_kernel_wave.hip)."""
    # ---- vector registers
    vOff16, vOff8, vTmp = V(0), V(1), V(2 + L)
    # a slot's event on diagonal d has index d - 1 - x; the events sit in LDS by index mod 256, mirrored, so the address is
    # a per-slot constant (LDS_EV + 16 * ((-1 - x) mod 256), set when the slot's k-mer enters) plus 16 * (d mod 256)
    vEvSlot = [V(2 + j) for j in range(L)]
    e0 = (3 + L + 1) // 2 * 2
    E = [V(e0 + 4 * j, 4) for j in range(L)]           # event (mean, noise) of the layer's cell
    PY = [V(e0 + 4 * L + 2 * j, 2) for j in range(L)]  # gap-Y emission
    c0 = e0 + 6 * L
    C = [[V(c0 + 36 * j + 2 * q, 2) for q in range(NCONST)] for j in range(L)]
    x0 = c0 + 36 * L
    # cells of a diagonal per parity of the diagonal and layer: m | pm | x | y  ((m, pm) and (x, y) leave as pairs)
    X = [[V(x0 + 8 * (L * p + j), 8) for j in range(L)] for p in range(2)]
    r0 = x0 + 16 * L
    R = [V(r0 + 6 * p, 6) for p in range(2)]           # layer L-1 of the lane below (m, x, y) per parity
    t0 = r0 + 12
    pool = Pool(t0, 255)
    k = Kernel(pool)

    def Xm(p, j): return X[p][j].sub(0, 2)
    def Xpm(p, j): return X[p][j].sub(2, 2)
    def Xx(p, j): return X[p][j].sub(4, 2)
    def Xy(p, j): return X[p][j].sub(6, 2)
    CMU, CSD, CRSD, CK1, CNMU, CNSD, CRNSD, CK2 = range(8)
    CPXO, CPXE = 16, 17

    # ---- scalar registers
    sArg, sWg = S(0, 2), S(2)
    sRing, sRow0, sRow1 = S(4, 2), S(6, 2), S(8, 2)
    sD, sTop, sXmin, sXmax, sRingMask, sDmod = S(10), S(11), S(12), S(13), S(14), S(15)
    sMask = [S(16 + 2 * j, 2) for j in range(L)]
    sInL, sInJ, sOutL, sOutJ = S(22), S(23), S(24), S(25)
    sEv, sLY, sEvHi = S(26, 2), S(28), S(29)
    sTrack, sLX, sStop = S(30, 2), S(32), S(33)
    sCtlPtr = S(34, 2)
    sCtl = S(36, 8)                                     # stepMin, stepMax, full, spare
    sStepMin, sStepMax, sFull = sCtl.sub(0, 2), sCtl.sub(2, 2), sCtl.sub(4, 2)
    sTMM, sTXM, sTYM, sTMY, sTYY = S(44, 2), S(46, 2), S(48, 2), S(50, 2), S(52, 2)
    s7p5, sNinf = S(54, 2), S(56, 2)
    sBit, sRet, sStagePC = S(58, 2), S(60, 2), S(62, 2)
    sT = [S(64 + i) for i in range(8)]
    sP = [S(64 + 2 * i, 2) for i in range(4)]           # the same as pairs
    sCtx, sState = S(72, 2), S(74, 2)
    sWin = S(76, 16)                                    # the window's plan record
    sClk0, sRt0 = S(92), S(93)                          # (low words: a launch is far shorter than 2^32 ticks)
    sMaskTab = S(94, 2)
    sK = S(96, 4)                                       # (prologue only)
    # what a ring row's stores cover (from the mask table): the band and the two slots next to it, which are parked here --
    # the sweep back loads a row under the same lanes and finds -inf emissions where a k-mer has just left or is about
    # to enter, which is what keeps cells outside the band out of its recurrence
    # (two sets, by the parity of the diagonal: a step asks for the next diagonal's while it works under its own)
    sMaskS = [[S(96 + 2 * j, 2) for j in range(L)], [S(80 + 2 * j, 2) for j in range(L)]]
    sMaskPy2 = [S(56, 2), S(86, 2)]                     # ... and the last layer's 8-byte emissions (16 lanes to a line)
    sFullM = S(86, 2)                                   # all ones on a diagonal whose gap states go to the ring too
    sCtxBytes = S(62)                                   # (prologue only: sStagePC is set after it)
    def W(f): return sWin.sub(f)
    sWindow = W(13)                                     # (a spare word of the plan record)
    sPlanRec = S(90, 2)                                 # (its last two words: the record's address, to read it again at the end)
    lbl = lambda s: ".L_%s_%s" % (name, s)

    def load_store_masks(q):
        """sMaskS[q] <- the lanes ring row sD is stored under"""
        k.salu("s_lshl_b32", sT[0], sD, 6)
        k.salu("s_add_u32", sT[0], sT[0], 32)
        k.smem("s_load_dwordx4", S(sMaskS[q][0].i, 4), sMaskTab, sT[0])
        k.salu("s_add_u32", sT[0], sT[0], 16)
        if PY2_X2:
            k.smem("s_load_dwordx2", sMaskS[q][2], sMaskTab, sT[0])
            k.salu("s_add_u32", sT[0], sT[0], 8)
            k.smem("s_load_dwordx2", sMaskPy2[q], sMaskTab, sT[0])
        else:
            k.smem("s_load_dwordx2", sMaskS[q][2], sMaskTab, sT[0])

    def masked_store(mask, dwords, voff, data, off, gate=None):
        if "NOSTORE" in ABLATE or ("NOSTORE4" in ABLATE and dwords == 4) or ("NOSTORE2" in ABLATE and dwords == 2):
            return
        if "STOREALL" in ABLATE:
            if gate is None:
                k.gstore(dwords, voff, data, sRow0 if off < 4096 else sRow1, off % 4096)
            return
        if isinstance(mask, tuple):
            k.salu("s_or_b64", "exec", mask[0], mask[1])
        elif gate is None:
            k.salu("s_mov_b64", "exec", mask)
        else:
            k.salu("s_and_b64", "exec", mask, gate)
        if "EXECONLY" not in ABLATE:
            k.gstore(dwords, voff, data, sRow0 if off < 4096 else sRow1, off % 4096)
        k.salu("s_mov_b64", "exec", -1)

    # ------------------------------------------------------------------ prologue
    k.label(name)
    # the forward wave goes first on its SIMD (CPECAN_ASM_ABLATE=NOPRIO: never; HALFPRIO: for the first half of every
    # step, which balanced the two sweeps of a window while their arithmetic was emitted chain after chain): now that the
    # sweep back is the shorter one by a fifth, this is what makes both last equally long -- the pass 4 % shorter than
    # with HALFPRIO, 7 % than with none (profiles/r03_same_box_ab_runs.txt)
    if "NOPRIO" not in ABLATE and "HALFPRIO" not in ABLATE:
        k.salu("s_setprio", 2)
    k.salu("s_memtime", sP[0])
    k.salu("s_memrealtime", sP[3])
    k.smem("s_load_dwordx4", sK, sArg, A_NITEMS)          # nItems, window, ringD, maxWindows
    k.smem("s_load_dwordx2", sP[1], sArg, A_PLANWIN)
    k.wait_lgkm()
    k.salu("s_mov_b32", sClk0, sT[0])
    k.salu("s_mov_b32", sRt0, sT[6])
    k.salu("s_cmp_ge_u32", sWg, sK.sub(0))
    k.branch("s_cbranch_scc1", lbl("exit"))
    k.salu("s_mov_b32", S(3), sK.sub(1))                 # window (sWindow lives in the plan record, loaded below)
    k.salu("s_sub_u32", sRingMask, sK.sub(2), 1)
    # the window's plan record: planWin + (wg * maxWindows + window) * 64
    k.salu("s_mul_i32", sT[0], sWg, sK.sub(3))
    k.salu("s_add_u32", sT[0], sT[0], S(3))
    k.salu("s_lshl_b32", sT[0], sT[0], 6)
    k.add64(sP[1], sP[1], sT[0])
    k.smem("s_load_dwordx16", sWin, sP[1], 0)
    k.smem("s_load_dwordx2", sP[2], sArg, A_ITEMS)
    k.wait_lgkm()
    k.salu("s_mov_b32", sWindow, S(3))
    k.salu("s_mov_b64", sPlanRec, sP[1])
    k.salu("s_cmp_ge_i32", sWindow, W(W_NWIN))
    k.branch("s_cbranch_scc1", lbl("exit"))
    # item record
    k.salu("s_lshl_b32", sT[0], sWg, 7)
    k.add64(sP[2], sP[2], sT[0])
    k.smem("s_load_dword", sLX, sP[2], I_LX)
    k.smem("s_load_dword", sLY, sP[2], I_LY)
    k.smem("s_load_dwordx2", sP[3], sP[2], I_YOFF)
    k.smem("s_load_dwordx2", sEv, sArg, A_EVENTS)
    k.smem("s_load_dwordx2", sP[0], sP[2], I_DIAGBASE)
    k.smem("s_load_dwordx2", sMaskTab, sArg, A_MASKTAB)
    k.wait_lgkm()
    k.salu("s_lshl_b64", sP[0], sP[0], 6)
    k.add64(sMaskTab, sMaskTab, sT[0], sT[1])
    k.smem("s_load_dword", sT[1], sP[2], I_MODEL)
    k.wait_lgkm()
    # events of this alignment: events + 24 * yOff
    k.salu("s_mul_i32", sT[2], sT[6], 24)
    k.salu("s_mul_hi_u32", sT[3], sT[6], 24)
    k.add64(sEv, sEv, sT[2], sT[3])
    # the model's transitions: models + model * stride * 8
    k.smem("s_load_dwordx2", sP[2], sArg, A_MODELS)
    k.smem("s_load_dwordx2", sP[3], sArg, A_MODELSTRIDE)
    k.wait_lgkm()
    k.salu("s_lshl_b32", sT[6], sT[6], 3)
    k.salu("s_mul_i32", sT[2], sT[1], sT[6])
    k.salu("s_mul_hi_u32", sT[3], sT[1], sT[6])
    k.add64(sP[2], sP[2], sT[2], sT[3])
    k.smem("s_load_dwordx2", sTMM, sP[2], 0 * 8)
    k.smem("s_load_dwordx2", sTXM, sP[2], 1 * 8)
    k.smem("s_load_dwordx2", sTYM, sP[2], 2 * 8)
    k.smem("s_load_dwordx2", sTMY, sP[2], 4 * 8)
    k.smem("s_load_dwordx2", sTYY, sP[2], 6 * 8)
    # track rows of this alignment: track + trackBase[wg] * 160
    k.smem("s_load_dwordx2", sP[3], sArg, A_TRACKBASE)
    k.smem("s_load_dwordx2", sTrack, sArg, A_TRACK)
    k.wait_lgkm()
    k.salu("s_lshl_b32", sT[0], sWg, 3)
    k.smem("s_load_dwordx2", sP[3], sP[3], sT[0])
    k.wait_lgkm()
    k.salu("s_mul_i32", sT[2], sT[6], TRACK_ROW_BYTES)
    k.salu("s_mul_hi_u32", sT[3], sT[6], TRACK_ROW_BYTES)
    k.add64(sTrack, sTrack, sT[2], sT[3])
    # ring of this alignment: ring + wg * ringDoubles * 8
    k.smem("s_load_dwordx4", S(64, 4), sArg, A_RING)      # ring, ringDoubles
    k.wait_lgkm()
    k.salu("s_lshl_b64", sP[1], sP[1], 3)
    k.salu("s_mul_i32", sT[4], sT[2], sWg)
    k.salu("s_mul_hi_u32", sT[5], sT[2], sWg)
    k.salu("s_mul_i32", sT[6], sT[3], sWg)
    k.salu("s_add_u32", sT[5], sT[5], sT[6])
    k.add64(sRing, sP[0], sT[4], sT[5])
    # control words: planCtl + planOff[wg] * 32
    k.smem("s_load_dwordx2", sP[3], sArg, A_PLANOFF)
    k.smem("s_load_dwordx2", sCtlPtr, sArg, A_PLANCTL)
    k.wait_lgkm()
    k.salu("s_lshl_b32", sT[0], sWg, 3)
    k.smem("s_load_dwordx2", sP[3], sP[3], sT[0])
    k.wait_lgkm()
    k.salu("s_lshl_b64", sP[3], sP[3], 5)
    k.add64(sCtlPtr, sCtlPtr, sT[6], sT[7])
    # state record and contexts
    k.smem("s_load_dwordx2", sState, sArg, A_STATES)
    k.smem("s_load_dwordx4", S(64, 4), sArg, A_CTX)       # ctx, ctxBytes (per context)
    k.wait_lgkm()
    k.salu("s_mul_i32", sT[4], sWg, STATE_BYTES)
    k.add64(sState, sState, sT[4])
    k.salu("s_mov_b32", sCtxBytes, sT[2])
    # three contexts per alignment: [0], [1] by window parity, [2] the start of an alignment (written by the host)
    k.salu("s_mul_i32", sT[4], sWg, 3)
    k.salu("s_mul_i32", sT[5], sT[4], sCtxBytes)
    k.salu("s_mul_hi_u32", sT[6], sT[4], sCtxBytes)
    k.add64(sCtx, sP[0], sT[5], sT[6])                    # context [0]
    # the one to load: window 0 -> [2], else [(window - 1) & 1]
    k.salu("s_add_u32", sT[0], sWindow, 1)
    k.salu("s_and_b32", sT[0], sT[0], 1)
    k.salu("s_cmp_eq_u32", sWindow, 0)
    k.salu("s_cselect_b32", sT[0], 2, sT[0])
    k.salu("s_mul_i32", sT[1], sT[0], sCtxBytes)
    k.add64(sP[3], sCtx, sT[1])                           # sP[3]: the context to load
    # ... and the one to save: [window & 1]
    k.salu("s_and_b32", sT[0], sWindow, 1)
    k.salu("s_mul_i32", sT[1], sT[0], sCtxBytes)
    k.add64(sCtx, sCtx, sT[1])

    # lane constants (v0 = lane on entry)
    k.valu("v_lshlrev_b32_e32", vOff8, 3, V(0))
    # the logAdd table: 64 doubles from the library's copy, one per lane
    k.smem("s_load_dwordx2", sP[1], sArg, A_COEF)
    k.wait_lgkm()
    tq = pool.take(2)
    k.gload(2, tq, vOff8, sP[1])
    k.valu("v_mov_b32_e32", vTmp, V(0))                   # (the lane)
    k.valu("v_lshlrev_b32_e32", vOff16, 4, V(0))          # (v0 is vOff16 from here on)
    k.ds_write(64, vOff8, tq, LDS_COEF)
    pool.give(tq)
    k.s_mov64_lit(s7p5, dbits(2.0 ** (47 if "OLDIDX" in ABLATE else 52)))
    k.magic = s7p5
    # every event of the LDS ring reads as (0, 0) until it is staged: a parked slot scores whatever its stale address
    # points at, and that has to be a number
    tz = pool.take(4)
    for q in range(4):
        k.valu("v_mov_b32_e32", tz.sub(q), 0)
    for q in range(((2 * EVN + L) * 16 + 1023) // 1024):
        if (q + 1) * 1024 <= (2 * EVN + L) * 16:
            k.ds_write(128, vOff16, tz, LDS_EV + q * 1024)
        else:
            k.valu("v_cmp_gt_u32_e32", "vcc", ((2 * EVN + L) * 16 - q * 1024) // 16, vTmp)
            k.salu("s_and_b64", "exec", "exec", "vcc")
            k.ds_write(128, vOff16, tz, LDS_EV + q * 1024)
            k.salu("s_mov_b64", "exec", -1)
    pool.give(tz)

    # ---- the context: constants and the cells of the last two diagonals
    for j in range(L):
        for q in range(9):
            k.gload(4, V(C[j][0].i + 4 * q, 4), vOff16, sP[3], (q % 4) * 1024)
            if q % 4 == 3 or q == 8:
                k.add64(sP[3], sP[3], 4096 if q % 4 == 3 else 1024)
    for p in range(2):
        for j in range(L):
            k.gload(2, Xm(p, j), vOff8, sP[3], 0)
            k.gload(4, V(Xx(p, j).i, 4), vOff16, sP[3], 512)
            k.add64(sP[3], sP[3], 1536)
    k.smem("s_load_dwordx8", S(16, 8), sP[3], 0)          # masks (6 dwords), in slot (lane, layer)
    k.smem("s_load_dwordx2", S(24, 2), sP[3], 32)         # out slot (lane, layer)
    k.wait_all()
    k.salu("s_mov_b32", sD, W(W_D0))                      # last diagonal done
    k.salu("s_mov_b32", sTop, W(W_TOP))
    k.salu("s_mov_b32", sXmin, W(W_XMIN0))
    k.salu("s_mov_b32", sXmax, W(W_XMAX0))
    # the k-mer a slot holds: the one of (xmax - P, xmax] that is congruent to the slot; its event address constant
    tq = pool.take(2)
    for j in range(L):
        k.valu("v_lshrrev_b32_e32", tq.lo, 4, vOff16)                # lane
        k.valu("v_mul_u32_u24_e32", tq.lo, L, tq.lo)
        k.valu("v_sub_u32_e32", tq.lo, P - j, tq.lo)                 # P - slot
        k.valu("v_add_u32_e32", tq.lo, sXmax, tq.lo)                 # n = xmax + P - slot  (> 0)
        k.valu("v_mov_b32_e32", tq.hi, "0x%x" % ((1 << 32) // P + 1))
        k.valu("v_mul_hi_u32", tq.hi, tq.lo, tq.hi)                  # n / P
        k.valu("v_mul_u32_u24_e32", tq.hi, P, tq.hi)
        k.valu("v_sub_u32_e32", tq.lo, tq.lo, tq.hi)                 # n mod P = xmax - x
        k.valu("v_subrev_u32_e32", tq.lo, sXmax, tq.lo)              # -x
        k.valu("v_add_u32_e32", tq.lo, -1, tq.lo)                    # -1 - x
        k.valu("v_and_b32_e32", tq.lo, EVN - 1, tq.lo)
        k.valu("v_lshlrev_b32_e32", tq.lo, 4, tq.lo)
        k.valu("v_add_u32_e32", vEvSlot[j], LDS_EV, tq.lo)
    pool.give(tq)
    # the rotated neighbours of both diagonals
    for p in range(2):
        for q, src in enumerate((Xm(p, L - 1), Xx(p, L - 1), Xy(p, L - 1))):
            k.ror64(R[p].sub(2 * q, 2), src)
    # stage from scratch everything the first diagonal's block can ask for
    k.salu("s_add_u32", sD, sD, 1)
    load_store_masks(0)
    load_store_masks(1)
    k.salu("s_sub_u32", sEvHi, sD, sXmax)
    k.salu("s_sub_u32", sEvHi, sEvHi, 2)
    k.pc_of(sStagePC, lbl("stage"))
    k.salu("s_swappc_b64", sRet, sStagePC)
    k.forget()
    # the first diagonal's events
    k.salu("s_and_b32", sDmod, sD, EVN - 1)
    k.salu("s_lshl_b32", sDmod, sDmod, 4)
    for j in range(L):
        k.valu("v_add_u32_e32", E[j].lo, sDmod, vEvSlot[j])
        k.ds_read(128, E[j], E[j].lo)
    k.salu("s_bitcmp1_b32", sD, 0)
    k.branch("s_cbranch_scc1", lbl("odd"))

    # ------------------------------------------------------------------ one anti-diagonal
    def step(p):
        """Diagonal sD of parity p: X[p] holds the diagonal before last and receives this one, X[1-p] the last one."""
        q = 1 - p
        k.mods = " nt" if "NT" in ABLATE else ""
        k.drain_lgkm()       # this diagonal's store masks (asked for a diagonal ago); the events are in too
        pend = []
        k.salu("s_add_u32", sD, sD, 1)
        if "NOMASKF" not in ABLATE:
            load_store_masks(q)  # the next diagonal's
        k.salu("s_sub_u32", sD, sD, 1)
        k.lgkm = []          # (waited for at the top of the next step)
        # band edges: the k-mer that leaves first (its slot is parked), then the one that enters
        k.salu("s_bitcmp1_b64", sStepMin, sD)
        if "NOEVENTS" not in ABLATE:
            k.branch("s_cbranch_scc1", lbl("leave%d" % p))
        k.label(lbl("left%d" % p))
        k.salu("s_bitcmp1_b64", sStepMax, sD)
        if "NOEVENTS" not in ABLATE:
            k.branch("s_cbranch_scc1", lbl("enter%d" % p))
        k.label(lbl("entered%d" % p))
        k.lgkm = pend
        # this diagonal's ring row
        k.salu("s_and_b32", sT[0], sD, sRingMask)
        k.salu("s_mul_i32", sT[0], sT[0], ROW_BYTES)
        k.add64(sRow0, sRing, sT[0])
        k.add64(sRow1, sRow0, 4096)

        def lower(j, w):   # (x-1, y) on the last diagonal: m (w = 0) or x (w = 1)
            return (Xm(q, j - 1), Xx(q, j - 1))[w] if j else R[q].sub(2 * w, 2)

        def middle(j, w):  # (x-1, y-1) on the diagonal before: m, x, y
            return (Xm(p, j - 1), Xx(p, j - 1), Xy(p, j - 1))[w] if j else R[p].sub(2 * w, 2)

        if "HALFPRIO" in ABLATE:
            k.salu("s_setprio", 2)
        # P1: gap X from the lower cell -- needs no emission
        aa, bb = [pool.take(2) for _ in range(L)], [pool.take(2) for _ in range(L)]
        for j in range(L):
            k.add(aa[j], lower(j, 0), C[j][CPXO])
        for j in range(L):
            k.add(bb[j], lower(j, 1), C[j][CPXE])
        recs = k.ladd_front_group(list(zip(aa, bb)), free=aa + bb)
        k.ladd_rows(recs)
        # P2: match emissions; the sums of the middle cell's gap-X state, which the gap-X results are about to overwrite
        k.gauss_group([(Xpm(p, j), E[j].sub(0, 2), C[j][CMU], C[j][CSD], C[j][CRSD], C[j][CK1]) for j in range(L)])
        gn = [pool.take(2) for _ in range(L)]
        k.gauss_group([(gn[j], E[j].sub(2, 2), C[j][CNMU], C[j][CNSD], C[j][CRNSD], C[j][CK2]) for j in range(L)])
        for j in range(L):
            k.add(Xpm(p, j), Xpm(p, j), gn[j])
        pool.give(*gn)
        bsum = [pool.take(2) for _ in range(L)]
        for j in range(L):
            k.add(bsum[j], Xpm(p, j), sTXM)
        for j in range(L):
            k.add(bsum[j], middle(j, 1), bsum[j])
        # P3: gap X done
        k.need_recs(recs)
        k.ladd_back_group(recs, [Xx(p, j) for j in range(L)])
        # P4: match from the middle cell, first two terms
        aa = [pool.take(2) for _ in range(L)]
        for j in range(L):
            k.add(aa[j], Xpm(p, j), sTMM)
        for j in range(L):
            k.add(aa[j], middle(j, 0), aa[j])
        recs = k.ladd_front_group(list(zip(aa, bsum)), free=aa + bsum)
        k.ladd_rows(recs)
        # P5: gap-Y emissions; the third match term (the middle cell's gap-Y state dies with this diagonal's gap-Y results)
        k.gauss_group([(PY[j], E[j].sub(0, 2), C[j][8 + CMU], C[j][8 + CSD], C[j][8 + CRSD], C[j][8 + CK1]) for j in range(L)])
        gn = [pool.take(2) for _ in range(L)]
        k.gauss_group([(gn[j], E[j].sub(2, 2), C[j][8 + CNMU], C[j][8 + CNSD], C[j][8 + CRNSD], C[j][8 + CK2])
                       for j in range(L)])
        for j in range(L):
            k.add(PY[j], PY[j], gn[j])
        pool.give(*gn)
        if "NOPY" not in ABLATE:                      # (a store between stretches of arithmetic: stores issued back to back hold the wave up)
            masked_store((sMaskS[p][0], sMaskS[p][1]), 4, vOff16, V(PY[0].i, 4), OFF_PY)
        csum = [pool.take(2) for _ in range(L)]
        for j in range(L):
            k.add(csum[j], Xpm(p, j), sTYM)
        for j in range(L):
            k.add(csum[j], middle(j, 2), csum[j])
        # (the last layer's leave with the two registers after them, which hold a constant: 16 bytes per lane again)
        if "NOPY" in ABLATE:
            pass
        elif PY2_X2:
            masked_store(sMaskPy2[p], 2, vOff8, PY[L - 1], OFF_PY + 1024)
        else:
            masked_store(sMaskS[p][L - 1], 4, vOff16, V(PY[L - 1].i, 4), OFF_PY + 1024)
        # P6
        k.need_recs(recs)
        k.ladd_back_group(recs, [Xm(p, j) for j in range(L)])
        if "HALFPRIO" in ABLATE:
            k.salu("s_setprio", 0)
        # P7: gap Y from the upper cell
        aa, bb = [pool.take(2) for _ in range(L)], [pool.take(2) for _ in range(L)]
        for j in range(L):
            k.add(aa[j], PY[j], sTMY)
        for j in range(L):
            k.add(bb[j], PY[j], sTYY)
        for j in range(L):
            k.add(aa[j], Xm(q, j), aa[j])
        for j in range(L):
            k.add(bb[j], Xy(q, j), bb[j])
        recs = k.ladd_front_group(list(zip(aa, bb)), free=aa + bb)
        k.ladd_rows(recs)
        # ... while its table reads are under way: the next diagonal's events, the gap-Y emissions' stores
        k.salu("s_add_u32", sDmod, sDmod, 16)
        k.salu("s_and_b32", sDmod, sDmod, 16 * (EVN - 1))
        for j in range(L):
            k.valu("v_add_u32_e32", E[j].lo, sDmod, vEvSlot[j])
            k.ds_read(128, E[j], E[j].lo)
        # P9 (on a diagonal the sweep back reads all three states of, the gap states leave too: out of line, one diagonal
        # in ten -- a store under an empty EXEC holds the wave up as long as any other)
        k.need_recs(recs)
        k.ladd_back_group(recs, [Xy(p, j) for j in range(L)])
        k.salu("s_bitcmp1_b64", sFull, sD)
        k.branch("s_cbranch_scc1", lbl("full%d" % p))
        k.label(lbl("fulled%d" % p))
        # P10: the match cell's third term
        recs = k.ladd_front_group([(Xm(p, j), csum[j]) for j in range(L)], free=csum)
        k.ladd_rows(recs)
        # ... meanwhile the gap states leave where the sweep back reads them again, and layer L-1 is rotated up a lane
        k.ror64(R[p].sub(2, 2), Xx(p, L - 1))
        k.ror64(R[p].sub(4, 2), Xy(p, L - 1))
        # P11: (Fm, pm) leave as they are finished; the match cells of layer L-1 go up a lane
        k.need_recs(recs)
        k.ladd_back_group(recs, [Xm(p, j) for j in range(L)])
        masked_store(sMaskS[p][0], 4, vOff16, V(Xm(p, 0).i, 4), 0)
        k.ror64(R[p].sub(0, 2), Xm(p, L - 1))
        for j in range(1, L):
            masked_store(sMaskS[p][j], 4, vOff16, V(Xm(p, j).i, 4), j * LAYER_BYTES)
        k.salu("s_add_u32", sD, sD, 1)

    k.label(lbl("even"))
    step(0)
    state_after = list(k.lgkm)
    k.salu("s_cmp_gt_i32", sD, sTop)
    k.branch("s_cbranch_scc1", lbl("done"))
    k.label(lbl("odd"))
    k.vm = []
    step(1)
    k.salu("s_cmp_le_i32", sD, sStop)
    k.branch("s_cbranch_scc1", lbl("even"))
    # a staging block ends here, or the window
    k.salu("s_cmp_gt_i32", sD, sTop)
    k.branch("s_cbranch_scc1", lbl("done"))
    k.salu("s_swappc_b64", sRet, sStagePC)
    k.branch("s_branch", lbl("even"))

    # ------------------------------------------------------------------ out of line: band edges, full rows
    for p in range(2):
        # a diagonal whose gap states go to the ring too
        k.forget()
        k.label(lbl("full%d" % p))
        for j in range(L):
            masked_store(sMaskS[p][j], 4, vOff16, V(X[p][j].i + 4, 4), j * LAYER_BYTES + OFF_FXY)
        k.branch("s_branch", lbl("fulled%d" % p))
    for p in range(2):
        # the k-mer at the band's low end leaves: its slot scores -inf from now on (K1, K2 of both tables, the gap-X sums)
        k.forget()
        k.label(lbl("leave%d" % p))
        k.salu("s_add_u32", sXmin, sXmin, 1)
        k.salu("s_lshl_b64", sBit, 1, sOutL)
        for j in range(L - 1):
            k.salu("s_cmp_eq_u32", sOutJ, j)
            k.branch("s_cbranch_scc1", lbl("leave%d_%d" % (p, j)))
        for j in reversed(range(L)):
            if j < L - 1:
                k.label(lbl("leave%d_%d" % (p, j)))
            k.salu("s_andn2_b64", sMask[j], sMask[j], sBit)
            k.salu("s_mov_b64", "exec", sBit)
            for q in (CK1, CK2, 8 + CK1, 8 + CK2, CPXO, CPXE):
                k.valu("v_mov_b32_e32", C[j][q].lo, 0)
                k.valu("v_mov_b32_e32", C[j][q].hi, "0xfff00000")
            k.salu("s_mov_b64", "exec", -1)
            if j < L - 1:
                k.salu("s_add_u32", sOutJ, sOutJ, 1)
            else:
                k.salu("s_mov_b32", sOutJ, 0)
                k.salu("s_add_u32", sOutL, sOutL, 1)
                k.salu("s_and_b32", sOutL, sOutL, 63)
            k.branch("s_branch", lbl("left%d" % p))
        # the next k-mer enters at the band's high end: its row comes from the LDS ring of rows
        k.forget()
        k.label(lbl("enter%d" % p))
        k.salu("s_add_u32", sXmax, sXmax, 1)
        k.salu("s_and_b32", sT[0], sXmax, ROWN - 1)
        k.salu("s_mul_i32", sT[0], sT[0], TRACK_ROW_BYTES)
        k.salu("s_add_u32", sT[0], sT[0], LDS_ROWS)
        k.valu("v_mov_b32_e32", vTmp, sT[0])
        k.salu("s_sub_u32", sT[1], -1, sXmax)
        k.salu("s_and_b32", sT[1], sT[1], EVN - 1)
        k.salu("s_lshl_b32", sT[1], sT[1], 4)
        k.salu("s_add_u32", sT[1], sT[1], LDS_EV)
        k.salu("s_lshl_b64", sBit, 1, sInL)
        for j in range(L - 1):
            k.salu("s_cmp_eq_u32", sInJ, j)
            k.branch("s_cbranch_scc1", lbl("enter%d_%d" % (p, j)))
        for j in reversed(range(L)):
            if j < L - 1:
                k.label(lbl("enter%d_%d" % (p, j)))
            k.salu("s_or_b64", sMask[j], sMask[j], sBit)
            k.salu("s_mov_b64", "exec", sBit)
            for q in range(9):
                k.ds_read(128, V(C[j][0].i + 4 * q, 4), vTmp, 16 * q)
            # ... and its event of this diagonal (the read the last step issued went by the slot's old constant)
            k.valu("v_mov_b32_e32", vEvSlot[j], sT[1])
            k.valu("v_add_u32_e32", vTmp, sDmod, vEvSlot[j])
            k.ds_read(128, E[j], vTmp)
            k.salu("s_mov_b64", "exec", -1)
            if j < L - 1:
                k.salu("s_add_u32", sInJ, sInJ, 1)
            else:
                k.salu("s_mov_b32", sInJ, 0)
                k.salu("s_add_u32", sInL, sInL, 1)
                k.salu("s_and_b32", sInL, sInL, 63)
            k.drain_lgkm()
            k.branch("s_branch", lbl("entered%d" % p))

    # ------------------------------------------------------------------ staging: control words, events, k-mer rows
    k.forget()
    k.label(lbl("stage"))
    # the control words of sD's block, and where the block (or the window) ends
    k.salu("s_lshr_b32", sT[0], sD, 6)
    k.salu("s_lshl_b32", sT[0], sT[0], 5)
    k.smem("s_load_dwordx8", sCtl, sCtlPtr, sT[0])
    k.salu("s_or_b32", sStop, sD, BLOCK - 1)
    k.salu("s_min_i32", sStop, sStop, sTop)
    # events up to index (block end + 1) - xmin - 1: the first diagonal of the next block reads before that block is staged
    k.salu("s_or_b32", sT[1], sD, BLOCK - 1)
    k.salu("s_add_u32", sT[1], sT[1], 2)
    k.salu("s_sub_u32", sT[1], sT[1], sXmin)              # sT[1]: first event index not needed yet
    tE = pool.take(4)
    tI, tA, tB = pool.take(2), pool.take(2), pool.take(2)
    k.label(lbl("stage_ev"))
    k.salu("s_cmp_ge_i32", sEvHi, sT[1])
    k.branch("s_cbranch_scc1", lbl("stage_rows"))
    k.valu("v_lshrrev_b32_e32", tI.lo, 4, vOff16)          # lane
    k.valu("v_add_u32_e32", tI.lo, sEvHi, tI.lo)           # e
    k.valu("v_mov_b32_e32", tE.sub(0), 0)
    k.valu("v_mov_b32_e32", tE.sub(1), 0)
    k.valu("v_mov_b32_e32", tE.sub(2), 0)
    k.valu("v_mov_b32_e32", tE.sub(3), 0)
    k.valu("v_cmp_le_i32_e32", "vcc", 0, tI.lo)
    k.valu("v_cmp_gt_i32_e64", sP[1], sLY, tI.lo)
    k.salu("s_and_b64", sP[1], sP[1], "vcc")
    k.valu("v_cmp_gt_i32_e64", sP[2], sT[1], tI.lo)        # e below the limit: these lanes write the ring
    k.salu("s_and_b64", sP[1], sP[1], sP[2])
    k.valu("v_mul_u32_u24_e32", tA.lo, 24, tI.lo)
    k.salu("s_mov_b64", "exec", sP[1])
    k.gload(4, tE, tA.lo, sEv)
    k.salu("s_mov_b64", "exec", sP[2])
    k.valu("v_and_b32_e32", tB.lo, EVN - 1, tI.lo)
    k.valu("v_lshlrev_b32_e32", tA.hi, 4, tB.lo)
    k.valu("v_add_u32_e32", tA.hi, LDS_EV, tA.hi)
    k.wait_vm()
    k.ds_write(128, tA.hi, tE, 0)
    k.ds_write(128, tA.hi, tE, 16 * EVN)
    k.valu("v_cmp_gt_u32_e32", "vcc", L, tB.lo)
    k.salu("s_and_b64", "exec", "exec", "vcc")
    k.ds_write(128, tA.hi, tE, 32 * EVN)
    k.salu("s_mov_b64", "exec", -1)
    k.salu("s_add_u32", sEvHi, sEvHi, 64)
    k.branch("s_branch", lbl("stage_ev"))
    k.label(lbl("stage_rows"))
    k.salu("s_mov_b32", sEvHi, sT[1])
    # rows of the k-mers xmax + 1 .. xmax + 64 (one lane each; past the last column: that column's, never installed)
    k.valu("v_lshrrev_b32_e32", tI.lo, 4, vOff16)
    k.valu("v_add_u32_e32", tI.lo, sXmax, tI.lo)
    k.valu("v_add_u32_e32", tI.lo, 1, tI.lo)
    k.valu("v_min_i32_e32", tI.hi, sLX, tI.lo)
    k.valu("v_mul_u32_u24_e32", tA.lo, TRACK_ROW_BYTES, tI.hi)
    k.valu("v_and_b32_e32", tB.lo, ROWN - 1, tI.lo)
    k.valu("v_mul_u32_u24_e32", tB.lo, TRACK_ROW_BYTES, tB.lo)
    k.valu("v_add_u32_e32", tB.lo, LDS_ROWS, tB.lo)
    rows = [pool.take(4) for _ in range(3)]
    # ... as many as k-mers enter during the block (the control word's bits, which the load above has brought by now) and
    # one to spare, not all 64: 144 bytes a lane
    k.drain_lgkm()
    k.salu("s_bcnt1_i32_b64", sT[2], sStepMax)
    k.salu("s_add_u32", sT[2], sT[2], 1)
    k.salu("s_bfm_b64", sP[2], sT[2], 0)                   # (a width of 6 bits: 63 lanes at most ...
    k.salu("s_cmp_gt_u32", sT[2], 63)                      # ... a k-mer entering on every diagonal of the block takes all 64)
    k.salu("s_cselect_b64", "exec", -1, sP[2])
    for base in range(0, 9, 3):
        for q in range(3):
            k.gload(4, rows[q], tA.lo, sTrack, 16 * (base + q))
        for q in range(3):
            k.ds_write(128, tB.lo, rows[q], 16 * (base + q))
    k.salu("s_mov_b64", "exec", -1)
    pool.give(tE, tI, tA, tB, *rows)
    k.wait_all()
    k.salu("s_setpc_b64", sRet)

    # ------------------------------------------------------------------ the window is swept: hand it to the sweep back
    k.forget()
    k.label(lbl("done"))
    k.wait_all()
    # an estimate of the window's totalProbability: the cells of the top diagonal dotted with the end vector the sweep
    # back starts from (stateMachine.c:1179-1207), in any order -- it steers the candidate test, the post kernel checks
    # every exact total against it.  log-sum-exp around the largest term, single precision past the subtraction.
    return k, locals()


def forward_tail(k, v):
    """(continued: split only to keep the functions readable)"""
    g = dict(v)
    pool, L_, name = g["pool"], L, g["name"]
    S_, lbl = S, g["lbl"]
    sArg, sWg, sT, sP, sWin, sState, sCtx = g["sArg"], g["sWg"], g["sT"], g["sP"], g["sWin"], g["sState"], g["sCtx"]
    sTop, sMask, sWindow, sNinf = g["sTop"], g["sMask"], g["sWindow"], g["sNinf"]
    sTMM, sTXM, sTYM = g["sTMM"], g["sTXM"], g["sTYM"]
    sInL, sInJ, sOutL, sOutJ, sClk0, sRt0 = g["sInL"], g["sInJ"], g["sOutL"], g["sOutJ"], g["sClk0"], g["sRt0"]
    Xm, Xx, Xy, C = g["Xm"], g["Xx"], g["Xy"], g["C"]
    vOff16, vOff8 = g["vOff16"], g["vOff8"]
    W = g["W"]
    # the plan record again (the loop used its registers)
    k.salu("s_mov_b32", sT[4], sWindow)
    k.salu("s_mov_b64", sP[3], g["sPlanRec"])
    k.smem("s_load_dwordx16", sWin, sP[3], 0)
    k.wait_lgkm()
    k.salu("s_mov_b32", sWindow, sT[4])
    # the model's record again (its transitions 3..6 for the ragged end vector): models + model * stride * 8
    k.smem("s_load_dwordx2", sP[3], sArg, A_ITEMS)
    k.wait_lgkm()
    k.salu("s_lshl_b32", sT[0], sWg, 7)
    k.add64(sP[3], sP[3], sT[0])
    k.smem("s_load_dword", sT[0], sP[3], I_RAGGEDR)
    k.smem("s_load_dword", sT[1], sP[3], I_MODEL)
    k.smem("s_load_dwordx2", sP[2], sArg, A_MODELS)
    k.smem("s_load_dwordx2", sP[3], sArg, A_MODELSTRIDE)
    k.wait_lgkm()
    k.salu("s_lshl_b32", sT[6], sT[6], 3)
    k.salu("s_mul_i32", sT[2], sT[1], sT[6])
    k.salu("s_mul_hi_u32", sT[3], sT[1], sT[6])
    k.add64(sP[2], sP[2], sT[2], sT[3])
    sOX, sOY, sEX, sEY = S_(36, 2), S_(38, 2), S_(40, 2), S_(42, 2)  # (the control words are done with)
    k.smem("s_load_dwordx2", sOX, sP[2], 3 * 8)
    k.smem("s_load_dwordx2", sOY, sP[2], 4 * 8)
    k.smem("s_load_dwordx2", sEX, sP[2], 5 * 8)
    k.smem("s_load_dwordx2", sEY, sP[2], 6 * 8)
    k.wait_lgkm()
    e0, e1, e2 = pool.take(2), pool.take(2), pool.take(2)
    # at the end of a ragged alignment: ((open X + open Y) / 2, extend X, extend Y); otherwise the transitions into match
    k.salu("s_cmp_lg_u32", W(W_ATEND), 0)
    k.salu("s_cselect_b32", sT[0], sT[0], 0)
    k.valu("v_mov_b32_e32", e0.lo, sOX.lo)
    k.valu("v_mov_b32_e32", e0.hi, sOX.hi)
    k.add(e0, e0, sOY)
    k.mul(e0, e0, "0.5")
    k.valu("v_mov_b32_e32", e1.lo, sEX.lo)
    k.valu("v_mov_b32_e32", e1.hi, sEX.hi)
    k.valu("v_mov_b32_e32", e2.lo, sEY.lo)
    k.valu("v_mov_b32_e32", e2.hi, sEY.hi)
    k.salu("s_cmp_lg_u32", sT[0], 0)
    k.salu("s_cselect_b64", "vcc", -1, 0)
    for e, s in ((e0, sTMM), (e1, sTXM), (e2, sTYM)):
        t = pool.take(2)
        k.valu("v_mov_b32_e32", t.lo, s.lo)
        k.valu("v_mov_b32_e32", t.hi, s.hi)
        k.valu("v_cndmask_b32_e32", e.lo, t.lo, e.lo, "vcc")
        k.valu("v_cndmask_b32_e32", e.hi, t.hi, e.hi, "vcc")
        pool.give(t)
    # the top diagonal's parity
    terms = [pool.take(2) for _ in range(3 * L_)]
    for p in range(2):
        k.salu("s_and_b32", sT[1], sTop, 1)
        k.salu("s_cmp_eq_u32", sT[1], p)
        k.branch("s_cbranch_scc0", lbl("est_p%d" % p))
        for j in range(L_):
            k.add(terms[3 * j + 0], Xm(p, j), e0)
            k.add(terms[3 * j + 1], Xx(p, j), e1)
            k.add(terms[3 * j + 2], Xy(p, j), e2)
        k.label(lbl("est_p%d" % p))
    pool.give(e0, e1, e2)
    vS = [g["E"][0].sub(q) for q in range(4)] + [g["E"][1].sub(0)]  # (the events' registers are free by now)
    # cells outside the band do not count
    for j in range(L_):
        for q in range(3):
            t = terms[3 * j + q]
            k.valu("v_mov_b32_e32", vS[0], "0xfff00000")
            k.valu("v_cndmask_b32_e64", t.hi, vS[0], t.hi, sMask[j])
            k.valu("v_cndmask_b32_e64", t.lo, 0, t.lo, sMask[j])
    mx = pool.take(2)
    k.valu("v_max_f64", mx, terms[0], terms[1])
    for t in terms[2:]:
        k.valu("v_max_f64", mx, mx, t)
    vM, vTmp, vTmp2, vLane4, vSum = vS
    k.valu("v_cvt_f32_f64_e32", vM, mx)
    k.valu("v_lshrrev_b32_e32", vLane4, 2, vOff16)
    k.butterfly("v_max_f32_e32", vM, vTmp, vLane4, vTmp2)
    k.valu("v_cvt_f64_f32_e32", mx, vM)
    k.valu("v_mov_b32_e32", vSum, 0)
    for t in terms:
        k.add(t, t, Neg(mx))
        k.valu("v_cvt_f32_f64_e32", vTmp, t)
        k.valu("v_mul_f32_e32", vTmp, "0x%x" % LOG2E_F32, vTmp)
        k.valu("v_exp_f32_e32", vTmp, vTmp)
        k.nop(0)
        k.valu("v_add_f32_e32", vSum, vSum, vTmp)
    k.butterfly("v_add_f32_e32", vSum, vTmp, vLane4, vTmp2)
    k.valu("v_log_f32_e32", vSum, vSum)
    k.nop(0)
    k.valu("v_mul_f32_e32", vSum, "0x%x" % LN2_F32, vSum)
    est = terms[0]
    k.valu("v_cvt_f64_f32_e32", est, vSum)
    k.add(est, est, mx)

    # ---- the context for the next launch (and for the sweep back of this window)
    k.salu("s_mov_b64", sP[3], sCtx)
    for j in range(L_):
        for q in range(9):
            k.gstore(4, vOff16, V(C[j][0].i + 4 * q, 4), sP[3], (q % 4) * 1024)
            if q % 4 == 3 or q == 8:
                k.add64(sP[3], sP[3], 4096 if q % 4 == 3 else 1024)
    for p in range(2):
        for j in range(L_):
            k.gstore(2, vOff8, Xm(p, j), sP[3], 0)
            k.gstore(4, vOff16, V(Xx(p, j).i, 4), sP[3], 512)
            k.add64(sP[3], sP[3], 1536)
    # scalars and records leave through lane 0
    k.salu("s_mov_b64", "exec", 1)
    q4 = [pool.take(4) for _ in range(3)]
    srcs = [sMask[0].lo, sMask[0].hi, sMask[1].lo, sMask[1].hi, sMask[2].lo, sMask[2].hi, sInL, sInJ, sOutL, sOutJ, 0, 0]
    for i, s in enumerate(srcs):
        k.valu("v_mov_b32_e32", q4[i // 4].sub(i % 4), s)
    vZ = pool.take(2)
    k.valu("v_mov_b32_e32", vZ.lo, 0)
    for i in range(3):
        k.gstore(4, vZ.lo, q4[i], sP[3], 16 * i)
    # the window for the sweep back: win[window & 3] = { valid 1, top, from, to, atEnd, 0, 0, 0, est }
    k.salu("s_and_b32", sT[0], sWindow, 3)
    k.salu("s_mul_i32", sT[0], sT[0], WIN_BYTES)
    k.salu("s_add_u32", sT[0], sT[0], ST_WIN)
    k.add64(sP[2], sState, sT[0])
    w4 = [pool.take(4) for _ in range(2)]
    for i, s in enumerate((1, W(W_TOP), W(W_FROM), W(W_TO), W(W_ATEND), 0, 0, 0)):
        k.valu("v_mov_b32_e32", w4[i // 4].sub(i % 4), s)
    k.gstore(4, vZ.lo, w4[0], sP[2], 0)
    k.gstore(4, vZ.lo, w4[1], sP[2], 16)
    k.gstore(2, vZ.lo, est, sP[2], 32)
    # state: d, tracedBackTo, finished; cells
    s3 = pool.take(4)
    k.valu("v_mov_b32_e32", s3.sub(0), W(W_TOP))
    k.valu("v_mov_b32_e32", s3.sub(1), W(W_FROM))
    k.valu("v_mov_b32_e32", s3.sub(2), W(W_ATEND))
    k.lines.append("\tglobal_store_dwordx3 %s, %s, %s" % (vZ.lo, V(s3.i, 3), sState))
    k.n += 1
    c2 = pool.take(2)
    k.valu("v_mov_b32_e32", c2.lo, W(W_CELLS))
    k.valu("v_mov_b32_e32", c2.hi, W(W_CELLS + 1))
    k.gstore(2, vZ.lo, c2, sState, ST_CELLS)
    # the clocks this sweep took (shader clock and the 100 MHz reference)
    clk = pool.take(4)
    k.gload(4, clk, vZ.lo, sState, ST_CLKS)
    k.salu("s_memtime", sP[0])
    k.salu("s_memrealtime", sP[1])
    k.drain_lgkm()
    k.salu("s_sub_u32", sT[0], sT[0], sClk0)
    k.salu("s_mov_b32", sT[1], 0)
    k.salu("s_sub_u32", sT[2], sT[2], sRt0)
    k.salu("s_mov_b32", sT[3], 0)
    d4 = pool.take(4)
    for i in range(4):
        k.valu("v_mov_b32_e32", d4.sub(i), sT[i])
    k.valu("v_add_co_u32_e32", clk.sub(0), "vcc", clk.sub(0), d4.sub(0))
    k.valu("v_addc_co_u32_e32", clk.sub(1), "vcc", clk.sub(1), d4.sub(1), "vcc")
    k.valu("v_add_co_u32_e32", clk.sub(2), "vcc", clk.sub(2), d4.sub(2))
    k.valu("v_addc_co_u32_e32", clk.sub(3), "vcc", clk.sub(3), d4.sub(3), "vcc")
    k.gstore(4, vZ.lo, clk, sState, ST_CLKS)
    k.label(lbl("exit"))
    k.salu("s_endpgm")
    return k


# ====================================================================================================== backward
# more of the formats: the scratch of one alignment (wv_backward_kernel() of cpecan_kernel_wave.hip), the mask table
MASK_BYTES = 64
MASK_GROUP = int(os.environ.get("CPECAN_ASM_MASK_GROUP", "8"))  # lanes a ring row's store / load masks are rounded to: 8 lanes x 16 bytes = one 128-byte line
WINTOTAL_BYTES = 32
CAND_PER_DIAG = 4
CAND_SLACK = 0.25


B_M0 = 8 + 4 * L + 4                      # backward kernel: first register of M[q][j] (2 * L * q + 2 * j on) ...
B_BX0 = B_M0 + 6 * L                      # ... of BX[j], then BY[j] (the emulator tests read the cells out of these)


def backward_kernel(name):
    """The sweep back of one traceback window, one wave per alignment (phase S of backward_window() of
    cpecan_kernel_wave.hip): the backward cells of every diagonal from the traceback point down, the decode candidates
    (cells whose F.match + B.match comes within CAND_SLACK of the posterior threshold against the forward kernel's
    estimate of the window's totalProbability) and, on the diagonals where the reference refreshes totalProbability,
    the backward operands of that sum parked for the post kernel.

    Everything that lives across diagonals rotates by the diagonal mod 3, so the loop body is written three times:
    M[q] the match cells of the diagonals = q, T[q] the ring's (Fm, pm) pairs of such a diagonal, PYB[q] its gap-Y
    emissions, PMB[q] its match emissions (copied out of T[q], whose registers take the row three diagonals down: a row is
    asked for two diagonals before the sweep reaches it), SM[q] its band (lanes per layer, first and last column) from the
    mask table.

    A ring row is loaded under the band of the row and of the row above it: the forward sweep stores -inf emissions for
    the slot of the k-mer that enters next, so a slot whose k-mer has just left the band (going down) reads -inf
    emissions on that diagonal, which is what takes its cells out of the recurrence; whatever a slot further outside
    holds (values of an earlier diagonal) only ever meets -inf cells."""
    vOff16, vOff8, vTmp, vTmp2 = V(0), V(1), V(2), V(3)
    vThr, vCthr = V(4, 2), V(6, 2)                     # candidate threshold in force / once decoding has begun
    PX = [V(8 + 4 * j, 4) for j in range(L)]           # (gap-X open sum, gap-X extend sum) of the slot's k-mer
    RP = V(8 + 4 * L, 4)                               # ... of layer 0 of the lane above
    m0 = B_M0
    M = [[V(m0 + 2 * L * q + 2 * j, 2) for j in range(L)] for q in range(3)]
    b0 = B_BX0
    BX = [V(b0 + 2 * j, 2) for j in range(L)]
    BY = [V(b0 + 2 * L + 2 * j, 2) for j in range(L)]
    UM = [V(b0 + 4 * L + 2 * j, 2) for j in range(L)]  # upper-block sums of the diagonal above: By + (py + tP)
    UY = [V(b0 + 6 * L + 2 * j, 2) for j in range(L)]
    t0 = b0 + 8 * L
    T = [[V(t0 + 4 * L * q + 4 * j, 4) for j in range(L)] for q in range(3)]
    p0 = t0 + 12 * L
    # gap-Y emissions as the ring holds them: (py0, py1), (py2, unused)
    pyw = 6 if PY2_X2 else 8
    PYB4 = [[V(p0 + pyw * q + 4 * h, 4 if h == 0 or not PY2_X2 else 2) for h in range(2)] for q in range(3)]
    PYB = [[V(p0 + pyw * q + 2 * j, 2) for j in range(L)] for q in range(3)]
    q0 = p0 + pyw * 3
    sLmPy2 = S(64, 2)                                   # the lanes the last layer's 8-byte emissions are loaded under
    PMB = [[V(q0 + 2 * L * q + 2 * j, 2) for j in range(L)] for q in range(3)]  # match emissions, kept two diagonals longer
    # F.match + B.match of the diagonal above, kept on the diagonal before a refresh of totalProbability: its second half
    FBS = V(q0 + 6 * L, 2 * L)
    pool0 = q0 + 8 * L
    pool = Pool(pool0, 255)
    k = Kernel(pool)
    def Tf(q, j): return T[q][j].sub(0, 2)
    def Tpm(q, j): return T[q][j].sub(2, 2)
    def PXo(j): return PX[j].sub(0, 2)
    def PXe(j): return PX[j].sub(2, 2)

    sArg, sWg, sWindow = S(0, 2), S(2), S(3)
    sRing0, sRing1 = S(4, 2), S(6, 2)
    sTd, sTo, sFrom, sTop, sTpost0, sRingMask, sNCand, sNTot = [S(8 + i) for i in range(8)]
    sCandCap, sRefCnt, sStop, sAtEnd = S(16), S(17), S(18), S(19)
    sRow0, sRow1 = S(20, 2), S(22, 2)
    SM = [S(24 + 8 * q, 8) for q in range(3)]
    def SMm(q, j): return SM[q].sub(2 * j, 2)
    def SMxmin(q): return SM[q].sub(6)
    def SMxmax(q): return SM[q].sub(7)
    sB = [S(24 + i) for i in range(24)]                # (the prologue's second scratch: the masks' registers)
    def sBp(i): return S(24 + i, 2)
    sMaskTab, sTrack = S(48, 2), S(50, 2)
    sTMM, sTXM, sTYM, sTMY, sTYY = S(52, 2), S(54, 2), S(56, 2), S(58, 2), S(60, 2)
    s7p5, sNinf = S(62, 2), S(64, 2)
    sRf, sWtot, sCandKx, sCandFb, sState = S(66, 2), S(68, 2), S(70, 2), S(72, 2), S(74, 2)
    sA = [S(76 + i) for i in range(16)]                # scratch; the window's plan record lands here first
    def sAp(i): return S(76 + i, 2)
    sRet, sStagePC = S(92, 2), S(94, 2)
    sLm = [S(96 + 2 * j, 2) for j in range(L)]         # the lanes a ring row is loaded under (mask table)
    sCmp = [S(84 + 2 * j, 2) for j in range(L)]        # candidate compares
    lbl = lambda s: ".L_%s_%s" % (name, s)

    # ------------------------------------------------------------------ prologue (common part)
    k.label(name)
    if "PRIOB" in ABLATE:
        k.salu("s_setprio", 2)
    k.smem("s_load_dwordx4", S(24, 4), sArg, A_NITEMS)     # sB0..3: nItems, window, ringD, maxWindows
    k.smem("s_load_dwordx2", sBp(4), sArg, A_PLANWIN)
    k.smem("s_load_dwordx2", sBp(6), sArg, A_STATES)
    k.smem("s_load_dwordx2", sBp(8), sArg, A_ITEMS)
    k.smem("s_load_dwordx2", sBp(10), sArg, A_LOGTHR)
    k.wait_lgkm()
    k.salu("s_cmp_ge_u32", sWg, sB[0])
    k.branch("s_cbranch_scc1", lbl("exit"))
    k.salu("s_mov_b32", sWindow, sB[1])
    k.salu("s_sub_u32", sRingMask, sB[2], 1)
    k.salu("s_mul_i32", sCandCap, sB[2], CAND_PER_DIAG * L)
    k.salu("s_mul_i32", sA[0], sWg, sB[3])
    k.salu("s_add_u32", sA[0], sA[0], sWindow)
    k.salu("s_lshl_b32", sA[0], sA[0], 6)
    k.add64(sBp(4), sBp(4), sA[0])
    k.smem("s_load_dwordx16", S(76, 16), sBp(4), 0)        # the plan record
    k.wait_lgkm()
    k.salu("s_cmp_ge_i32", sWindow, sA[W_NWIN])
    k.branch("s_cbranch_scc1", lbl("exit"))
    k.salu("s_mov_b32", sTop, sA[W_TOP])
    k.salu("s_mov_b32", sFrom, sA[W_FROM])
    k.salu("s_mov_b32", sTo, sA[W_TO])
    k.salu("s_mov_b32", sAtEnd, sA[W_ATEND])
    k.salu("s_mov_b32", sTpost0, sA[W_TPOST0])
    # the state record; the window as the forward kernel described it
    k.salu("s_mul_i32", sA[0], sWg, STATE_BYTES)
    k.add64(sState, sBp(6), sA[0])
    k.salu("s_and_b32", sA[0], sWindow, 3)
    k.salu("s_mul_i32", sA[0], sA[0], WIN_BYTES)
    k.salu("s_add_u32", sA[0], sA[0], ST_WIN)
    k.add64(sAp(2), sState, sA[0])
    k.smem("s_load_dword", sA[0], sAp(2), 0)               # valid
    k.smem("s_load_dwordx2", sAp(4), sAp(2), 32)           # est
    k.salu("s_lshl_b32", sA[1], sWg, 7)
    k.add64(sBp(8), sBp(8), sA[1])                         # the item
    k.smem("s_load_dword", sA[6], sBp(8), I_MODEL)
    k.smem("s_load_dword", sA[7], sBp(8), I_RAGGEDR)
    k.smem("s_load_dwordx2", sAp(8), sBp(8), I_DIAGBASE)
    k.smem("s_load_dwordx2", sAp(10), sArg, A_MODELS)
    k.smem("s_load_dwordx2", sAp(12), sArg, A_MODELSTRIDE)
    k.smem("s_load_dwordx2", sMaskTab, sArg, A_MASKTAB)
    k.wait_lgkm()
    k.salu("s_cmp_lg_u32", sA[0], 1)
    k.branch("s_cbranch_scc1", lbl("exit"))
    # candidates: F.match + B.match >= est + (log threshold - margin) - slack (a threshold of 0: -inf, every cell; the
    # lists overflow and the post kernel hands the window to the re-sweep kernel)
    k.valu("v_mov_b32_e32", vCthr.lo, sB[10])
    k.valu("v_mov_b32_e32", vCthr.hi, sB[11])
    k.s_mov64_lit(sBp(12), dbits(-CAND_SLACK))
    k.add(vCthr, vCthr, sBp(12))
    k.add(vCthr, vCthr, sAp(4))
    # the band table of this alignment; the model's transitions
    k.salu("s_lshl_b64", sAp(8), sAp(8), 6)
    k.add64(sMaskTab, sMaskTab, sA[8], sA[9])
    k.salu("s_lshl_b32", sA[12], sA[12], 3)
    k.salu("s_mul_i32", sA[0], sA[6], sA[12])
    k.salu("s_mul_hi_u32", sA[1], sA[6], sA[12])
    k.add64(sAp(10), sAp(10), sA[0], sA[1])
    k.smem("s_load_dwordx2", sTMM, sAp(10), 0 * 8)
    k.smem("s_load_dwordx2", sTXM, sAp(10), 1 * 8)
    k.smem("s_load_dwordx2", sTYM, sAp(10), 2 * 8)
    k.smem("s_load_dwordx2", sTMY, sAp(10), 4 * 8)
    k.smem("s_load_dwordx2", sTYY, sAp(10), 6 * 8)
    k.smem("s_load_dwordx2", sBp(12), sAp(10), 3 * 8)      # open X
    k.smem("s_load_dwordx2", sBp(14), sAp(10), 5 * 8)      # extend X
    k.smem("s_load_dwordx2", sBp(16), sArg, A_TRACKBASE)
    k.smem("s_load_dwordx2", sTrack, sArg, A_TRACK)
    k.wait_lgkm()
    # the end vector the sweep starts from (stateMachine.c:1179-1207): at the end of a ragged alignment
    # ((open X + open Y) / 2, extend X, extend Y), otherwise the transitions into match
    e = [pool.take(2) for _ in range(3)]
    k.salu("s_cmp_lg_u32", sAtEnd, 0)
    k.salu("s_cselect_b32", sA[7], sA[7], 0)
    k.valu("v_mov_b32_e32", e[0].lo, sB[12])
    k.valu("v_mov_b32_e32", e[0].hi, sB[13])
    k.add(e[0], e[0], sTMY)
    k.mul(e[0], e[0], "0.5")
    k.valu("v_mov_b32_e32", e[1].lo, sB[14])
    k.valu("v_mov_b32_e32", e[1].hi, sB[15])
    k.valu("v_mov_b32_e32", e[2].lo, sTYY.lo)
    k.valu("v_mov_b32_e32", e[2].hi, sTYY.hi)
    k.salu("s_cmp_lg_u32", sA[7], 0)
    k.salu("s_cselect_b64", "vcc", -1, 0)
    for ee, s in ((e[0], sTMM), (e[1], sTXM), (e[2], sTYM)):
        k.valu("v_mov_b32_e32", vTmp, s.lo)
        k.valu("v_mov_b32_e32", vTmp2, s.hi)
        k.valu("v_cndmask_b32_e32", ee.lo, vTmp, ee.lo, "vcc")
        k.valu("v_cndmask_b32_e32", ee.hi, vTmp2, ee.hi, "vcc")
    # track rows of this alignment (the gap-X sums of the k-mers that enter on the way down)
    k.salu("s_lshl_b32", sA[0], sWg, 3)
    k.smem("s_load_dwordx2", sBp(16), sBp(16), sA[0])
    k.smem("s_load_dwordx4", S(76 + 4, 4), sArg, A_RING)   # sA4..7: ring, ringDoubles
    k.smem("s_load_dwordx4", S(76 + 8, 4), sArg, A_CTX)    # sA8..11: ctx, ctxBytes
    k.smem("s_load_dwordx4", S(76 + 12, 4), sArg, A_SCRATCH)  # sA12..15: scratch, scratchBytes
    k.wait_lgkm()
    k.salu("s_mul_i32", sA[0], sB[16], TRACK_ROW_BYTES)
    k.salu("s_mul_hi_u32", sA[1], sB[16], TRACK_ROW_BYTES)
    k.add64(sTrack, sTrack, sA[0], sA[1])
    # ring of this alignment
    k.salu("s_lshl_b64", sAp(6), sAp(6), 3)
    k.salu("s_mul_i32", sA[0], sA[6], sWg)
    k.salu("s_mul_hi_u32", sA[1], sA[6], sWg)
    k.salu("s_mul_i32", sA[2], sA[7], sWg)
    k.salu("s_add_u32", sA[1], sA[1], sA[2])
    k.add64(sRing0, sAp(4), sA[0], sA[1])
    k.add64(sRing1, sRing0, 4096)
    # the context the forward kernel saved at this traceback point: [window & 1]
    k.salu("s_mul_i32", sA[0], sWg, 3)
    k.salu("s_and_b32", sA[1], sWindow, 1)
    k.salu("s_add_u32", sA[0], sA[0], sA[1])
    k.salu("s_mul_i32", sA[1], sA[0], sA[10])
    k.salu("s_mul_hi_u32", sA[2], sA[0], sA[10])
    k.add64(sAp(8), sAp(8), sA[1], sA[2])                  # sA8:9: the context
    # scratch of this alignment: [hit offsets | window totals | their terms | parked operands | hit masks | candidates]
    k.salu("s_mul_i32", sA[0], sA[14], sWg)
    k.salu("s_mul_hi_u32", sA[1], sA[14], sWg)
    k.salu("s_mul_i32", sA[2], sA[15], sWg)
    k.salu("s_add_u32", sA[1], sA[1], sA[2])
    k.add64(sAp(12), sAp(12), sA[0], sA[1])                # sc
    k.salu("s_lshl_b32", sA[0], sB[2], 3)                  # 2 * ringD * 4
    k.add64(sWtot, sAp(12), sA[0])
    k.salu("s_mul_hi_u32", sA[0], sB[2], "0xcccccccd")
    k.salu("s_lshr_b32", sA[0], sA[0], 3)
    k.salu("s_add_u32", sA[0], sA[0], 8)                   # nW = ringD / 10 + 8
    k.salu("s_mul_i32", sA[1], sA[0], WINTOTAL_BYTES)
    k.add64(sRf, sWtot, sA[1])                             # the refreshes' terms
    k.salu("s_mul_i32", sA[1], sA[0], 7 * P * 8)           # (them, and the operands the compiled sweep parks)
    k.add64(sCandKx, sRf, sA[1])                           # (the hit masks)
    k.salu("s_lshl_b32", sA[1], sB[2], 5)                  # 4 * ringD * 8
    k.add64(sCandKx, sCandKx, sA[1])
    k.salu("s_mul_i32", sA[1], sB[2], L * CAND_PER_DIAG * 8)
    k.add64(sCandFb, sCandKx, sA[1])
    # lane constants, the logAdd table, constants
    k.valu("v_lshlrev_b32_e32", vOff8, 3, V(0))
    k.smem("s_load_dwordx2", sAp(0), sArg, A_COEF)
    k.wait_lgkm()
    tq = pool.take(2)
    k.gload(2, tq, vOff8, sAp(0))
    k.valu("v_lshlrev_b32_e32", vOff16, 4, V(0))
    k.ds_write(64, vOff8, tq, LDS_COEF)
    pool.give(tq)
    k.s_mov64_lit(s7p5, dbits(2.0 ** (47 if "OLDIDX" in ABLATE else 52)))
    k.magic = s7p5
    # the gap-X sums of the slots at the traceback point are the forward wave's (a parked slot holds -inf)
    k.add64(sAp(8), sAp(8), 8 * 1024)
    for j in range(L):
        k.gload(4, PX[j], vOff16, sAp(8), 0)
        if j < L - 1:
            k.add64(sAp(8), sAp(8), 9 * 1024)
    # every cell and every ring value starts as -inf
    for r in range(m0, pool0, 2):
        k.valu("v_mov_b32_e32", V(r), 0)
        k.valu("v_mov_b32_e32", V(r + 1), "0xfff00000")
    k.salu("s_mov_b32", sNCand, 0)
    k.salu("s_mov_b32", sNTot, 0)
    k.salu("s_mov_b32", sTd, sTop)
    # decoding starts at tracedBackFrom: until then no cell is a candidate; the first refresh falls on the first
    # decoded diagonal
    k.valu("v_mov_b32_e32", vThr.lo, 0)
    k.valu("v_mov_b32_e32", vThr.hi, "0x7ff00000")
    k.salu("s_sub_u32", sRefCnt, sTop, sTpost0)
    k.salu("s_cmp_le_i32", sTop, sFrom)
    k.branch("s_cbranch_scc0", lbl("nodecode"))
    k.valu("v_mov_b32_e32", vThr.lo, vCthr.lo)
    k.valu("v_mov_b32_e32", vThr.hi, vCthr.hi)
    k.label(lbl("nodecode"))
    k.pc_of(sStagePC, lbl("stage"))
    k.wait_all()
    for h in range(2):
        k.rol64(RP.sub(2 * h, 2), PX[0].sub(2 * h, 2))
    def prefetch_terms(dreg, masks=None):
        """loads to LDS of what the refresh on diagonal dreg reads of the ring beyond the sweep's own loads: (Fx, Fy) of
        its row.  masks: the lanes of a ring row at most eight diagonals above, per layer -- the band moves by at most a
        column a diagonal, a lane per three columns, so those lanes and the eight below them (cyclically) cover the row
        wanted; None: every lane (a lane outside the band reads whatever the ring holds, nobody uses it)"""
        k.salu("s_and_b32", sA[0], dreg, sRingMask)
        k.salu("s_mul_i32", sA[0], sA[0], ROW_BYTES)
        k.add64(sAp(4), sRing0, sA[0])
        for j in range(L):
            if masks is not None:
                k.salu("s_lshr_b64", sAp(8), masks[j], 8)
                k.salu("s_lshl_b64", sAp(10), masks[j], 56)
                k.salu("s_or_b64", sAp(8), sAp(8), sAp(10))
                k.salu("s_or_b64", "exec", sAp(8), masks[j])
            k.add64(sAp(2), sAp(4), j * LAYER_BYTES + OFF_FXY)
            k.salu("s_mov_b32", "m0", LDS_RF_XY_T + j * 1024)
            k.nop(1)
            k.gload_lds(vOff16, sAp(2))
        if masks is not None:
            k.salu("s_mov_b64", "exec", -1)

    prefetch_terms(sTpost0)
    # which third of the loop the traceback point falls in
    k.salu("s_mul_hi_u32", sA[0], sTop, "0x55555556")
    k.salu("s_mul_i32", sA[0], sA[0], 3)
    k.salu("s_sub_u32", sA[0], sTop, sA[0])                # top mod 3
    k.salu("s_cmp_eq_u32", sA[0], 1)
    k.branch("s_cbranch_scc1", lbl("entry1"))
    k.salu("s_cmp_eq_u32", sA[0], 2)
    k.branch("s_cbranch_scc1", lbl("entry2"))

    def row_bases(dreg):
        """sRow0 / sRow1 <- byte address of ring row dreg (and + 4096)"""
        k.salu("s_and_b32", sA[0], dreg, sRingMask)
        k.salu("s_mul_i32", sA[0], sA[0], ROW_BYTES)
        k.add64(sRow0, sRing0, sA[0])
        k.add64(sRow1, sRing1, sA[0])

    def fetch_row(q, masks):
        """ring row at sRow0 -> T[q], PYB[q], each layer under its exec mask"""
        for j in range(L):
            k.salu("s_mov_b64", "exec", masks[j])
            off = j * LAYER_BYTES
            k.gload(4, T[q][j], vOff16, sRow0 if off < 4096 else sRow1, off % 4096)
            if j:
                if j == 1:
                    k.salu("s_or_b64", "exec", masks[0], masks[1])
                off = OFF_PY + (j - 1) * 1024
                if j == 2 and PY2_X2:
                    k.salu("s_mov_b64", "exec", sLmPy2)
                    k.gload(2, PYB4[q][1], vOff8, sRow0 if off < 4096 else sRow1, off % 4096)
                else:
                    k.gload(4, PYB4[q][j - 1], vOff16, sRow0 if off < 4096 else sRow1, off % 4096)
        k.salu("s_mov_b64", "exec", -1)

    def load_masks(q, dreg):
        """SM[q] <- band of diagonal dreg (clamped at 0) from the mask table"""
        k.salu("s_max_i32", sA[0], dreg, 0)
        k.salu("s_lshl_b32", sA[0], sA[0], 6)
        k.smem("s_load_dwordx8", SM[q], sMaskTab, sA[0])

    def load_row_masks(dreg):
        """sLm <- the lanes ring row dreg is loaded under"""
        k.salu("s_max_i32", sA[0], dreg, 0)
        k.salu("s_lshl_b32", sA[0], sA[0], 6)
        k.salu("s_add_u32", sA[0], sA[0], 32)
        k.smem("s_load_dwordx4", S(96, 4), sMaskTab, sA[0])
        k.salu("s_add_u32", sA[0], sA[0], 16)
        k.smem("s_load_dwordx2", S(100, 2), sMaskTab, sA[0])
        if PY2_X2:
            k.salu("s_add_u32", sA[0], sA[0], 8)
            k.smem("s_load_dwordx2", sLmPy2, sMaskTab, sA[0])

    # ------------------------------------------------------------------ entry: the traceback point = kk mod 3
    for kk in (0, 1, 2):
        k1, k2 = (kk + 1) % 3, (kk + 2) % 3
        k.forget()
        k.label(lbl("entry%d" % kk))
        load_masks(kk, sTd)
        k.salu("s_sub_u32", sA[1], sTd, 1)
        load_masks(k2, sA[1])
        k.salu("s_sub_u32", sA[1], sTd, 2)
        load_masks(k1, sA[1])
        load_row_masks(sTd)
        k.drain_lgkm()
        # the cells of the traceback point: the end vector wherever the band has a cell
        for j in range(L):
            k.salu("s_mov_b64", "exec", SMm(kk, j))
            for dst, src in ((M[kk][j], e[0]), (BX[j], e[1]), (BY[j], e[2])):
                k.valu("v_mov_b32_e32", dst.lo, src.lo)
                k.valu("v_mov_b32_e32", dst.hi, src.hi)
        k.salu("s_mov_b64", "exec", -1)
        # its ring row, and the two rows below it
        row_bases(sTd)
        fetch_row(kk, sLm)
        for back, q in ((1, k2), (2, k1)):
            k.salu("s_sub_u32", sA[1], sTd, back)
            load_row_masks(sA[1])
            k.drain_lgkm()
            row_bases(sA[1])
            fetch_row(q, sLm)
        k.salu("s_sub_u32", sA[1], sTd, 3)
        load_row_masks(sA[1])
        k.label(lbl("entered%d" % kk))
        k.salu("s_mov_b64", S(90, 2), SM[kk].sub(6, 2))
        k.salu("s_mov_b32", sA[2], SMxmin(kk))
        k.salu("s_swappc_b64", sRet, sStagePC)
        k.wait_all()
        k.branch("s_branch", lbl("tail%d" % kk))
    pool.give(*e)
    return k, locals()


def backward_loop(k, v):
    g = dict(v)
    k.mods = " nt" if "NT" in ABLATE or "NTB" in ABLATE else ""
    name, pool, lbl = g["name"], g["pool"], g["lbl"]
    M, BX, BY, UM, UY, T, PYB, PX, RP = g["M"], g["BX"], g["BY"], g["UM"], g["UY"], g["T"], g["PYB"], g["PX"], g["RP"]
    PMB, PYB4, sLmPy2, FBS = g["PMB"], g["PYB4"], g["sLmPy2"], g["FBS"]
    Tf, Tpm, PXo, PXe = g["Tf"], g["Tpm"], g["PXo"], g["PXe"]
    vOff16, vOff8, vTmp, vTmp2, vThr, vCthr = g["vOff16"], g["vOff8"], g["vTmp"], g["vTmp2"], g["vThr"], g["vCthr"]
    SM, SMm, SMxmin, SMxmax, sLm, sA, sAp = g["SM"], g["SMm"], g["SMxmin"], g["SMxmax"], g["sLm"], g["sA"], g["sAp"]
    sTd, sTo, sFrom, sTop, sTpost0, sNCand, sNTot, sCandCap, sRefCnt, sStop = \
        g["sTd"], g["sTo"], g["sFrom"], g["sTop"], g["sTpost0"], g["sNCand"], g["sNTot"], g["sCandCap"], g["sRefCnt"], g["sStop"]
    sTMM, sTXM, sTYM, sTMY, sTYY, s7p5, sNinf = g["sTMM"], g["sTXM"], g["sTYM"], g["sTMY"], g["sTYY"], g["s7p5"], g["sNinf"]
    sRow0, sRow1, sRf, sWtot, sCandKx, sCandFb, sState = g["sRow0"], g["sRow1"], g["sRf"], g["sWtot"], g["sCandKx"], g["sCandFb"], g["sState"]
    sRet, sStagePC, sTrack, sWindow, sMaskTab = g["sRet"], g["sStagePC"], g["sTrack"], g["sWindow"], g["sMaskTab"]
    row_bases, fetch_row, load_masks, load_row_masks, sCmp = g["row_bases"], g["fetch_row"], g["load_masks"], g["load_row_masks"], g["sCmp"]
    prefetch_terms, sRingMask, sRing0 = g["prefetch_terms"], g["sRingMask"], g["sRing0"]
    sNxt = S(90, 2)                            # band (first, last column) of the diagonal above the current one

    for kk in (0, 1, 2):
        k1, k2 = (kk + 1) % 3, (kk + 2) % 3   # the diagonals t + 1 and t + 2 (= t - 1) mod 3
        # ------------------------------------------------------------ head: the cells of diagonal t (t = kk mod 3)
        k.forget()
        k.label(lbl("head%d" % kk))
        k.drain_lgkm()                         # (the band of t - 1, asked for a diagonal ago)
        # of slot + 1: B.match and match emission of t + 2 (middle block), B.gapX of t + 1 with its k-mer's gap-X sums
        # (lower block); layer L-1 takes them from layer 0 of the lane above
        rhB, rhP, rBx = pool.take(2), pool.take(2), pool.take(2)
        k.rol64(rhB, M[k2][0])
        k.rol64(rhP, PMB[k2][0])
        k.rol64(rBx, BX[0])
        # gather form of cell_calculateBackward (:378-389): (t + 2) middle block first; stage by stage over the layers
        sB_ = [M[k2][j + 1] if j < L - 1 else rhB for j in range(L)]
        sP_ = [PMB[k2][j + 1] if j < L - 1 else rhP for j in range(L)]
        sBx_ = [BX[j + 1] if j < L - 1 else rBx for j in range(L)]
        pe = [PXe(j + 1) if j < L - 1 else RP.sub(2, 2) for j in range(L)]
        po = [PXo(j + 1) if j < L - 1 else RP.sub(0, 2) for j in range(L)]
        bmin, bxin, byin, y1, y2 = [[pool.take(2) for _ in range(L)] for _ in range(5)]
        for dst, t in ((bmin, sTMM), (bxin, sTXM), (byin, sTYM)):
            for j in range(L):
                k.add(dst[j], sP_[j], t)
        for j in range(L):
            k.add(y1[j], sBx_[j], pe[j])
        for j in range(L):
            k.add(y2[j], sBx_[j], po[j])
        for dst in (bmin, bxin, byin):
            for j in range(L):
                k.add(dst[j], sB_[j], dst[j])
        pool.give(rhB, rhP, rBx)
        # the k-mers that left / entered the band coming down to t: their slots change hands now, for the diagonals below
        k.salu("s_cmp_lg_u64", SM[k1].sub(6, 2), SM[kk].sub(6, 2))
        if "NOEVENTS" not in ABLATE:
            k.branch("s_cbranch_scc1", lbl("band%d" % kk))
        k.label(lbl("banded%d" % kk))
        # the band of t - 2 (into the registers of t + 1's, whose columns the tail still wants), then the ring row of t - 1
        k.salu("s_mov_b64", sNxt, SM[k1].sub(6, 2))
        k.salu("s_sub_u32", sA[1], sTd, 2)
        if "NOMASKB" not in ABLATE:
            load_masks(k1, sA[1])
        k.salu("s_sub_u32", sA[1], sTd, 2)
        row_bases(sA[1])
        # (t + 1, same slot) upper block, then (t + 1, slot + 1) lower block -- the reference's scatter order per state;
        # the loads of ring row t - 2 (into the registers of row t + 1) go out one layer at a time between stretches of
        # arithmetic
        ra = k.ladd_front_group([(bmin[j], UM[j]) for j in range(L)], free=bmin)
        rb = k.ladd_front_group([(byin[j], UY[j]) for j in range(L)], free=byin)
        k.ladd_rows(ra + rb)
        for j in range(L):
            k.salu("s_mov_b64", "exec", sLm[j])
            off = j * LAYER_BYTES
            if "NOLOAD" not in ABLATE:
                k.gload(4, T[k1][j], vOff16, sRow0 if off < 4096 else sRow1, off % 4096)
                if j and "NOPY" not in ABLATE:
                    if j == 1:
                        k.salu("s_or_b64", "exec", sLm[0], sLm[1])
                    off = OFF_PY + (j - 1) * 1024
                    if j == 2 and PY2_X2:
                        k.salu("s_mov_b64", "exec", sLmPy2)
                        k.gload(2, PYB4[k1][1], vOff8, sRow0 if off < 4096 else sRow1, off % 4096)
                    else:
                        k.gload(4, PYB4[k1][j - 1], vOff16, sRow0 if off < 4096 else sRow1, off % 4096)
            k.salu("s_mov_b64", "exec", -1)
        k.salu("s_sub_u32", sA[1], sTd, 3)
        if "NOMASKB" not in ABLATE:
            load_row_masks(sA[1])
        k.need_recs(ra)
        k.ladd_back_group(ra, [M[kk][j] for j in range(L)])
        rc = k.ladd_front_group([(bxin[j], y1[j]) for j in range(L)], free=bxin + y1)
        k.ladd_rows(rc)
        k.need_recs(rb)
        k.ladd_back_group(rb, BY)
        rd = k.ladd_front_group([(M[kk][j], y2[j]) for j in range(L)], free=y2)
        k.ladd_rows(rd)
        k.need_recs(rc)
        k.ladd_back_group(rc, BX)
        k.need_recs(rd)
        k.ladd_back_group(rd, [M[kk][j] for j in range(L)])
        # ------------------------------------------------------------ tail: what diagonal t hands down, its candidates
        k.label(lbl("tail%d" % kk))
        k.raw_wait_vm(2 * L if "NOPY" in ABLATE else 2 * (L + 2))   # this diagonal's ring row (the L + 2 loads of each of the next two may still be under way)
        for j in range(L):
            k.valu("v_mov_b32_e32", PMB[kk][j].lo, Tpm(kk, j).lo)
            k.valu("v_mov_b32_e32", PMB[kk][j].hi, Tpm(kk, j).hi)
        if "NOPY0" in ABLATE:
            for j in range(L):
                k.valu("v_mov_b32_e32", PYB[kk][j].lo, Tpm(kk, j).lo)
                k.valu("v_mov_b32_e32", PYB[kk][j].hi, Tpm(kk, j).hi)
        elif "NOPY" in ABLATE:                   # timing only: the work of computing the gap-Y emissions here
            for j in range(L):
                cs = [pool.take(4) for _ in range(4)]
                ev = pool.take(4)
                g0, g1, g2 = pool.take(2), pool.take(2), pool.take(2)
                k.valu("v_add_u32_e32", vTmp, 64 * j, vOff16)
                for i in range(4):
                    k.ds_read(128, cs[i], vOff16, offset=1024 * i)
                k.ds_read(128, ev, vTmp)
                k.gauss(PYB[kk][j], ev.sub(0, 2), cs[0].sub(0, 2), cs[0].sub(2, 2), cs[1].sub(0, 2), cs[1].sub(2, 2), g0, g1)
                k.gauss(g2, ev.sub(2, 2), cs[2].sub(0, 2), cs[2].sub(2, 2), cs[3].sub(0, 2), cs[3].sub(2, 2), g0, g1)
                k.add(PYB[kk][j], PYB[kk][j], g2)
                k.mul(PYB[kk][j], Tpm(kk, j), "4.0")                   # (benign: 16 x the match emissions instead)
                k.mul(PYB[kk][j], PYB[kk][j], "4.0")
                pool.give(*cs, ev, g0, g1, g2)
        for j in range(L):
            k.add(UM[j], PYB[kk][j], sTMY)
        for j in range(L):
            k.add(UY[j], PYB[kk][j], sTYY)
        for j in range(L):
            k.add(UM[j], BY[j], UM[j])
        for j in range(L):
            k.add(UY[j], BY[j], UY[j])
        fb = [pool.take(2) for _ in range(L)]
        for j in range(L):
            k.add(fb[j], Tf(kk, j), M[kk][j])
        for j in range(L):
            k.valu("v_cmp_ge_f64_e64", sCmp[j], fb[j], vThr)
        k.salu("s_or_b64", sAp(2), sCmp[0], sCmp[1])
        for j in range(2, L):
            k.salu("s_or_b64", sAp(2), sAp(2), sCmp[j])
        if "NOCAND" not in ABLATE:
            k.branch("s_cbranch_scc1", lbl("cand%d" % kk))
        k.label(lbl("canded%d" % kk))
        k.salu("s_sub_u32", sRefCnt, sRefCnt, 1)
        k.salu("s_cmp_lt_i32", sRefCnt, 1)     # 0: the diagonal below has a refresh of totalProbability; -1: this one
        k.branch("s_cbranch_scc1", lbl("refresh%d" % kk))
        k.label(lbl("refreshed%d" % kk))
        pool.give(*fb)
        k.salu("s_sub_u32", sTd, sTd, 1)
        k.salu("s_cmp_gt_i32", sTd, sStop)
        k.branch("s_cbranch_scc1", lbl("head%d" % k2))
        k.salu("s_mov_b32", sA[3], k2)
        k.salu("s_mov_b32", sA[2], SMxmin(k2))
        k.branch("s_branch", lbl("pause"))
        g["fb%d" % kk] = fb

    # ------------------------------------------------------------------ out of line
    for kk in (0, 1, 2):
        k1, k2 = (kk + 1) % 3, (kk + 2) % 3
        fb = g["fb%d" % kk]
        # band edges between t + 1 and t: the slot of a k-mer that left (at the band's high end) is parked, the one of the
        # k-mer that entered (at the low end) gets its gap-X sums from the LDS ring of rows
        k.forget()
        k.label(lbl("band%d" % kk))
        k.salu("s_and_b32", sA[0], SMxmin(kk), PXN - 1)
        k.salu("s_lshl_b32", sA[0], sA[0], 4)
        k.salu("s_add_u32", sA[0], sA[0], LDS_PX)
        k.valu("v_mov_b32_e32", vTmp, sA[0])
        for j in range(L):
            k.salu("s_andn2_b64", "exec", SMm(k1, j), SMm(kk, j))
            for q in range(2):
                k.valu("v_mov_b32_e32", PX[j].sub(2 * q), 0)
                k.valu("v_mov_b32_e32", PX[j].sub(2 * q + 1), "0xfff00000")
            k.salu("s_andn2_b64", "exec", SMm(kk, j), SMm(k1, j))
            k.ds_read(128, PX[j], vTmp)
        k.salu("s_mov_b64", "exec", -1)
        k.drain_lgkm()
        for h in range(2):
            k.rol64(RP.sub(2 * h, 2), PX[0].sub(2 * h, 2))
        k.branch("s_branch", lbl("banded%d" % kk))
        # decode candidates of diagonal t: (how far down the window, k-mer) and F.match + B.match, appended to the lists
        k.forget()
        k.label(lbl("cand%d" % kk))
        k.salu("s_sub_u32", sA[4], sTpost0, sTd)            # kPost
        # slot of the band's first k-mer: xmin mod P
        k.salu("s_mul_hi_u32", sA[5], SMxmin(kk), "0x%x" % ((1 << 32) // P + 1))
        k.salu("s_mul_i32", sA[5], sA[5], P)
        k.salu("s_sub_u32", sA[5], SMxmin(kk), sA[5])
        pool.hold(*fb)
        ci, cx = pool.take(2), pool.take(2)
        for j in range(L):
            skip = lbl("cand%d_%d" % (kk, j))
            k.salu("s_cmp_eq_u64", sCmp[j], 0)
            k.branch("s_cbranch_scc1", skip)
            k.valu("v_mbcnt_lo_u32_b32", ci.lo, sCmp[j].lo, 0)
            k.valu("v_mbcnt_hi_u32_b32", ci.lo, sCmp[j].hi, ci.lo)
            k.valu("v_add_u32_e32", ci.lo, sNCand, ci.lo)
            # the k-mer of this slot: xmin + ((slot - slot of xmin) mod P)
            k.valu("v_lshrrev_b32_e32", cx.lo, 4, vOff16)
            k.valu("v_mul_u32_u24_e32", cx.lo, L, cx.lo)
            k.valu("v_add_u32_e32", cx.lo, j, cx.lo)
            k.valu("v_subrev_u32_e32", cx.lo, sA[5], cx.lo)
            k.valu("v_add_u32_e32", cx.hi, P, cx.lo)
            k.valu("v_cmp_gt_i32_e32", "vcc", 0, cx.lo)
            k.valu("v_cndmask_b32_e32", cx.hi, cx.lo, cx.hi, "vcc")
            k.valu("v_add_u32_e32", cx.hi, SMxmin(kk), cx.hi)
            k.valu("v_mov_b32_e32", cx.lo, sA[4])
            k.valu("v_cmp_gt_u32_e32", "vcc", sCandCap, ci.lo)
            k.salu("s_and_b64", "exec", sCmp[j], "vcc")
            k.valu("v_lshlrev_b32_e32", ci.lo, 3, ci.lo)
            k.gstore(2, ci.lo, cx, sCandKx)
            k.gstore(2, ci.lo, fb[j], sCandFb)
            k.salu("s_mov_b64", "exec", -1)
            k.salu("s_bcnt1_i32_b64", sA[6], sCmp[j])
            k.salu("s_add_u32", sNCand, sNCand, sA[6])
            k.label(skip)
        pool.give(ci, cx, *fb)
        k.branch("s_branch", lbl("canded%d" % kk))
        # a refresh of totalProbability (:956-966) on diagonal t: the backward operands of its terms are parked -- B of this
        # diagonal, B.match and the match emission of the one above -- with the diagonal and the two bands
        k.forget()
        k.label(lbl("refresh%d" % kk))
        pool.hold(*fb)
        # the diagonal before a refresh: F.match + B.match of this one is the refresh's second half.  (Per cell of t + 1,
        # diagonalCalculationTotalProbability, :736-754, sums the match transitions from the cell below-left on t - 1
        # into it, times B.match(t + 1): the sum over the three states of t - 1 with the cell's match emission is, term
        # for term and in the same order, the forward recurrence of the match state, :365-375 -- F.match(t + 1) itself.)
        k.salu("s_cmp_eq_u32", sRefCnt, 0)
        k.branch("s_cbranch_scc0", lbl("refresh%d_now" % kk))
        for j in range(L):
            k.valu("v_mov_b32_e32", FBS.sub(2 * j), fb[j].lo)
            k.valu("v_mov_b32_e32", FBS.sub(2 * j + 1), fb[j].hi)
        k.branch("s_branch", lbl("refreshed%d" % kk))
        k.label(lbl("refresh%d_now" % kk))
        k.salu("s_mov_b32", sRefCnt, 9)
        # the refresh's first half: per cell of t, v = F(t) . B(t) -- (Fx, Fy) of t were fetched to LDS a refresh ago
        vPrev = pool.take(2)
        k.valu("v_add_u32_e32", vPrev.hi, vOff16, vOff8)       # 24 * lane: a lane's three terms lie together
        xyT = [pool.take(4) for _ in range(L)]
        for j in range(L):
            k.ds_read(128, xyT[j], vOff16, LDS_RF_XY_T + j * 1024)
        out = [pool.take(3 * 2), FBS]
        a1, a2 = [pool.take(2) for _ in range(L)], [pool.take(2) for _ in range(L)]
        for j in range(L):
            k.add(a1[j], xyT[j].sub(0, 2), BX[j])
            k.add(a2[j], xyT[j].sub(2, 2), BY[j])
        pool.give(*xyT)
        r = k.ladd_front_group([(fb[j], a1[j]) for j in range(L)])
        k.ladd_rows(r)
        k.need_recs(r)
        k.ladd_back_group(r, a1)
        r = k.ladd_front_group([(a1[j], a2[j]) for j in range(L)])
        k.ladd_rows(r)
        k.need_recs(r)
        k.ladd_back_group(r, [out[0].sub(2 * j, 2) for j in range(L)])
        pool.give(*a1)
        pool.give(*a2)
        k.salu("s_mul_i32", sA[0], sNTot, 2 * P * 8)
        k.add64(sAp(4), sRf, sA[0])
        for f in range(2):
            if "NOSTORE" not in ABLATE:
                k.gstore(4, vPrev.hi, out[f].sub(0, 4), sAp(4), f * P * 8)
                k.gstore(2, vPrev.hi, out[f].sub(4, 2), sAp(4), f * P * 8 + 16)
        pool.give(vPrev, out[0])
        pool.give(*fb)
        pool.hold(*fb)
        rec = [pool.take(4), pool.take(4)]
        k.salu("s_mov_b64", "exec", 1)
        k.salu("s_add_u32", sA[1], sTd, 1)
        k.salu("s_cmp_le_i32", sA[1], sTop)
        k.salu("s_cselect_b32", sA[1], 1, 0)
        for i, s in enumerate((sTd, SMxmin(kk), SMxmax(kk), sNxt.lo, sNxt.hi, sA[1], 0, "0xfff00000")):
            k.valu("v_mov_b32_e32", rec[i // 4].sub(i % 4), s)
        k.valu("v_mov_b32_e32", vTmp, 0)
        k.salu("s_lshl_b32", sA[0], sNTot, 5)
        k.add64(sAp(4), sWtot, sA[0])
        k.gstore(4, vTmp, rec[0], sAp(4), 0)
        k.gstore(4, vTmp, rec[1], sAp(4), 16)
        k.salu("s_mov_b64", "exec", -1)
        k.salu("s_add_u32", sNTot, sNTot, 1)
        pool.give(*rec)
        pool.give(*fb)
        k.salu("s_sub_u32", sA[12], sTd, 10)
        prefetch_terms(sA[12], sLm)
        k.branch("s_branch", lbl("refreshed%d" % kk))

    # ------------------------------------------------------------------ the loop pauses: end of the window, the diagonal
    # where decoding begins, or a staging block of gap-X rows
    k.forget()
    k.label(lbl("pause"))
    k.salu("s_cmp_le_i32", sTd, sTo)
    k.branch("s_cbranch_scc1", lbl("done"))
    k.salu("s_cmp_lg_u32", sTd, sFrom)
    k.branch("s_cbranch_scc1", lbl("pause_stage"))
    k.valu("v_mov_b32_e32", vThr.lo, vCthr.lo)
    k.valu("v_mov_b32_e32", vThr.hi, vCthr.hi)
    k.label(lbl("pause_stage"))
    k.salu("s_swappc_b64", sRet, sStagePC)
    for q in (1, 2):
        k.salu("s_cmp_eq_u32", sA[3], q)
        k.branch("s_cbranch_scc1", lbl("head%d" % q))
    k.branch("s_branch", lbl("head0"))

    # ------------------------------------------------------------------ staging: gap-X rows of the k-mers that can enter
    # before the next pause; where that is.  In: sTd the next diagonal to compute, sA[2] the band's first column there
    k.forget()
    k.label(lbl("stage"))
    tx, ta = pool.take(2), pool.take(4)
    k.valu("v_lshrrev_b32_e32", tx.lo, 4, vOff16)
    k.valu("v_sub_u32_e32", tx.lo, sA[2], tx.lo)           # x = xmin - lane: the k-mer that may enter at the very next diagonal, and on
    k.valu("v_cmp_le_i32_e32", "vcc", 0, tx.lo)
    k.valu("v_mul_u32_u24_e32", tx.hi, TRACK_ROW_BYTES, tx.lo)
    k.salu("s_and_b64", "exec", "exec", "vcc")
    k.gload(4, ta, tx.hi, sTrack, 16 * 8)
    k.valu("v_and_b32_e32", tx.lo, PXN - 1, tx.lo)
    k.valu("v_lshlrev_b32_e32", tx.lo, 4, tx.lo)
    k.ds_write(128, tx.lo, ta, LDS_PX)
    k.salu("s_mov_b64", "exec", -1)
    pool.give(tx, ta)
    # the next pause: 64 diagonals on, the diagonal decoding begins at, or the end of the window
    k.salu("s_sub_u32", sStop, sTd, BLOCK)
    k.salu("s_cmp_gt_i32", sTd, sFrom)
    k.salu("s_cselect_b32", sA[0], sFrom, sTo)
    k.salu("s_max_i32", sStop, sStop, sA[0])
    k.salu("s_max_i32", sStop, sStop, sTo)
    k.wait_all()
    k.salu("s_setpc_b64", sRet)

    # ------------------------------------------------------------------ the window is swept: counts for the post kernel
    k.forget()
    k.label(lbl("done"))
    k.salu("s_and_b32", sA[0], sWindow, 3)
    k.salu("s_mul_i32", sA[0], sA[0], WIN_BYTES)
    k.salu("s_add_u32", sA[0], sA[0], ST_WIN)
    k.add64(sAp(2), sState, sA[0])
    k.salu("s_mov_b64", "exec", 1)
    k.valu("v_mov_b32_e32", vTmp, 0)
    r2 = pool.take(2)
    k.valu("v_mov_b32_e32", r2.lo, sNCand)
    k.valu("v_mov_b32_e32", r2.hi, sNTot)
    k.gstore(2, vTmp, r2, sAp(2), 20)                      # nCand, nRefresh
    k.valu("v_mov_b32_e32", vTmp2, 3)
    k.gstore(1, vTmp, vTmp2, sAp(2), 0)                    # valid = 3: swept back, the post kernel's turn
    pool.give(r2)
    k.label(lbl("exit"))
    k.salu("s_endpgm")
    return k


KERNEL_TEMPLATE = """\t.text
\t.protected\t{name}
\t.globl\t{name}
\t.p2align\t8
\t.type\t{name},@function
{body}
\t.section\t.rodata,"a",@progbits
\t.p2align\t6, 0x0
\t.amdhsa_kernel {name}
\t\t.amdhsa_group_segment_fixed_size {lds}
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size {kernarg}
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_dispatch_ptr 0
\t\t.amdhsa_user_sgpr_queue_ptr 0
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_user_sgpr_dispatch_id 0
\t\t.amdhsa_user_sgpr_kernarg_preload_length 0
\t\t.amdhsa_user_sgpr_kernarg_preload_offset 0
\t\t.amdhsa_user_sgpr_private_segment_size 0
\t\t.amdhsa_uses_dynamic_stack 0
\t\t.amdhsa_enable_private_segment 0
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_sgpr_workgroup_id_y 0
\t\t.amdhsa_system_sgpr_workgroup_id_z 0
\t\t.amdhsa_system_sgpr_workgroup_info 0
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr {vgprs}
\t\t.amdhsa_next_free_sgpr {sgprs}
\t\t.amdhsa_accum_offset {accum}
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_round_mode_32 0
\t\t.amdhsa_float_round_mode_16_64 0
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t\t.amdhsa_fp16_overflow 0
\t\t.amdhsa_tg_split 0
\t\t.amdhsa_exception_fp_ieee_invalid_op 0
\t\t.amdhsa_exception_fp_denorm_src 0
\t\t.amdhsa_exception_fp_ieee_div_zero 0
\t\t.amdhsa_exception_fp_ieee_overflow 0
\t\t.amdhsa_exception_fp_ieee_underflow 0
\t\t.amdhsa_exception_fp_ieee_inexact 0
\t\t.amdhsa_exception_int_div_zero 0
\t.end_amdhsa_kernel
\t.text
.Lend_{name}:
\t.size\t{name}, .Lend_{name}-{name}
"""

META_TEMPLATE = """  - .agpr_count:     0
    .args:
      - .offset:         0
        .size:           {kernarg}
        .value_kind:     by_value
    .group_segment_fixed_size: {lds}
    .kernarg_segment_align: 8
    .kernarg_segment_size: {kernarg}
    .language:       OpenCL C
    .language_version:
      - 2
      - 0
    .max_flat_workgroup_size: 64
    .name:           {name}
    .private_segment_fixed_size: 0
    .sgpr_count:     {sgprs}
    .sgpr_spill_count: 0
    .symbol:         {name}.kd
    .uniform_work_group_size: 1
    .uses_dynamic_stack: false
    .vgpr_count:     {vgprs}
    .vgpr_spill_count: 0
    .wavefront_size: 64
"""


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
    kernels = []
    name = "cpecan_k_asm_forward_l%d" % L
    k, v = forward_kernel(name)
    forward_tail(k, v)
    kernels.append(dict(name=name, body=k.text(), lds=LDS_F_BYTES, kernarg=ARGS_BYTES, vgprs=256, sgprs=102, accum=256,
                        stats=k.stats))
    name = "cpecan_k_asm_backward_l%d" % L
    k, v = backward_kernel(name)
    backward_loop(k, v)
    nv = (k.pool.first + 2 * k.pool.high + 7) // 8 * 8
    kernels.append(dict(name=name, body=k.text(), lds=LDS_B_BYTES, kernarg=ARGS_BYTES, vgprs=nv, sgprs=102, accum=nv,
                        stats=dict(k.stats, vgprs=nv)))
    text = ['\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', "\t.amdhsa_code_object_version 6"]
    for kk in kernels:
        text.append(KERNEL_TEMPLATE.format(**kk))
    text.append("\t.amdgpu_metadata\n---\namdhsa.kernels:")
    for kk in kernels:
        text.append(META_TEMPLATE.format(**kk).rstrip("\n"))
    text.append("amdhsa.target:   amdgcn-amd-amdhsa--gfx950\namdhsa.version:\n  - 1\n  - 2\n...\n\n\t.end_amdgpu_metadata")
    open(out, "w").write("\n".join(text) + "\n")
    if len(sys.argv) > 2:
        defs = dict(ASM_L=L, ASM_ROW_BYTES=ROW_BYTES, ASM_LAYER_BYTES=LAYER_BYTES, ASM_OFF_FXY=OFF_FXY, ASM_OFF_PY=OFF_PY, ASM_PY2_X2=int(PY2_X2), ASM_CTX_X=CTX_X, ASM_CTX_S=CTX_S, ASM_CTX_BYTES=CTX_BYTES,
                    ASM_MAX_WIDTH=MAX_WIDTH, ASM_PLANWIN_BYTES=PLANWIN_BYTES, ASM_CTL_BYTES=CTL_BYTES, ASM_BLOCK=BLOCK,
                    ASM_ARGS_BYTES=ARGS_BYTES, ASM_NCONST=NCONST, ASM_MASK_BYTES=MASK_BYTES, ASM_MASK_GROUP=MASK_GROUP, ASM_LDS_F_BYTES=LDS_F_BYTES, ASM_LDS_B_BYTES=LDS_B_BYTES)
        with open(sys.argv[2], "w") as h:
            h.write("/* generated by asm/gen_sweeps.py: what the assembly sweeps and the C++ side agree on */\n")
            h.write("#ifndef CPECAN_ASM_GEN_H_\n#define CPECAN_ASM_GEN_H_\n")
            for kname, val in defs.items():
                h.write("#define %s %d\n" % (kname, val))
            h.write("#endif\n")
    for kk in kernels:
        sys.stderr.write("%s: %s\n" % (kk["name"], kk["stats"]))


if __name__ == "__main__":
    main()

/*
 * cpecan_kernel_general5.hip -- banded forward / backward / posterior DP for the reference's
 * 5-state symbol machine (stateMachine5, impl/stateMachine.c:829-865; BASELINE configs[0]: DNA
 * against DNA, the reference's own CPU-runnable case).
 *
 * Same structure as cpecan_kernel_general.hip: one 256-thread workgroup per work item (= one
 * getPosteriorProbsWithBanding call, impl/pairwiseAligner.c:870-1006), threads stride over the
 * cells of an anti-diagonal, forward diagonals in HBM ([cell][state], :567), three rotating
 * backward diagonals in a small workspace, any band width (the un-anchored 1 kb x 1 kb case is a
 * full matrix).  Posterior decode only.
 *
 * States (inc/stateMachine.h:31-35): match 0, shortGapX 1, shortGapY 2, longGapX 3, longGapY 4.
 * Backward is a gather with the reference's scatter order kept per target state: the cell on d+2
 * (its middle block), then the cell (d+1, xmy-1) (its upper block, transitions in listed order),
 * then (d+1, xmy+1) (its lower block).
 */
#include "cpecan_device.h"

/* transition slots: the order of struct _StateMachine5 (inc/stateMachine.h:108-124) */
enum {
    T5_MATCH_CONTINUE = 0, T5_MATCH_FROM_SHORT_GAP_X, T5_MATCH_FROM_LONG_GAP_X, T5_GAP_SHORT_OPEN_X,
    T5_GAP_SHORT_EXTEND_X, T5_GAP_SHORT_SWITCH_TO_X, T5_GAP_LONG_OPEN_X, T5_GAP_LONG_EXTEND_X,
    T5_GAP_LONG_SWITCH_TO_X, T5_MATCH_FROM_SHORT_GAP_Y, T5_MATCH_FROM_LONG_GAP_Y,
    T5_GAP_SHORT_OPEN_Y, T5_GAP_SHORT_EXTEND_Y, T5_GAP_SHORT_SWITCH_TO_Y, T5_GAP_LONG_OPEN_Y,
    T5_GAP_LONG_EXTEND_Y, T5_GAP_LONG_SWITCH_TO_Y
};
#define S5 5

namespace {

struct Ctx5 {
    const int *L, *R;
    const long long *pre;
    const char *cx, *cy;    /* nucleotides of this item */
    const double *t;        /* 17 transitions */
    const double *mm;       /* 4 x 4 match emissions */
    const double *gx, *gy;  /* 4 + 4 gap emissions */
    double *F, *Bws;
    int maxWidth;
    double *ldsF; /* the forward cells of the last three diagonals, [d % 3][cell][state]; NULL: read them from HBM */
    int ldsW;
};

/* emissions_discrete_getBaseIndex impl/stateMachine.c:104-118: anything but upper-case ACGT is "not a
 * base" (4097 there); index < 0 is the "n" sentinel of sequence_getBase (:308-312) */
__device__ __forceinline__ int base_of(const char *s, long long i) {
    if (i < 0) return 4;
    const char ch = s[i];
    return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
}
/* emissions_symbol_getGapProb / getMatchProb :155-173 (N-free input is a precondition, quirk Q3) */
__device__ __forceinline__ double e_gap(const double *g, int i) { return i < 4 ? g[i] : CP_NEG_INF; }
__device__ __forceinline__ double e_match(const double *m, int ix, int iy) {
    return ix < 4 && iy < 4 ? m[ix * 4 + iy] : CP_NEG_INF;
}

__device__ __forceinline__ const double *fcell5(const Ctx5 &c, long long d, int xmy) {
    if (d < 0) return nullptr;
    const int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    return c.F + (c.pre[d] + ((xmy - l) >> 1)) * S5;
}
/* the same for the forward sweep's own neighbours (diagonals d - 1 and d - 2 of the diagonal being computed) */
__device__ __forceinline__ const double *fcell5_sweep(const Ctx5 &c, long long d, int xmy) {
    if (d < 0) return nullptr;
    const int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    if (c.ldsF) return c.ldsF + ((d % 3) * (long long) c.ldsW + ((xmy - l) >> 1)) * S5;
    return c.F + (c.pre[d] + ((xmy - l) >> 1)) * S5;
}
__device__ __forceinline__ double *bslot5(const Ctx5 &c, long long d) {
    return c.Bws + (d % 3) * (long long) c.maxWidth * S5;
}
__device__ __forceinline__ const double *bcell5(const Ctx5 &c, long long d, long long dTop, int xmy) {
    if (d > dTop) return nullptr;
    const int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    return bslot5(c, d) + ((xmy - l) >> 1) * S5;
}

/* match state reached from the five states of `middle` (stateMachine5_cellCalculate :843-851) */
__device__ __forceinline__ double match_from(const double *middle, double eP, const double *t) {
    double m = CP_NEG_INF;
    m = cp_logAdd(m, middle[0] + (eP + t[T5_MATCH_CONTINUE]));
    m = cp_logAdd(m, middle[1] + (eP + t[T5_MATCH_FROM_SHORT_GAP_X]));
    m = cp_logAdd(m, middle[2] + (eP + t[T5_MATCH_FROM_SHORT_GAP_Y]));
    m = cp_logAdd(m, middle[3] + (eP + t[T5_MATCH_FROM_LONG_GAP_X]));
    m = cp_logAdd(m, middle[4] + (eP + t[T5_MATCH_FROM_LONG_GAP_Y]));
    return m;
}

/* cell_calculateForward (:365-376) over stateMachine5_cellCalculate (:829-865) */
__device__ __forceinline__ void forward_cell5(const Ctx5 &c, long long d, int xmy, double o[S5]) {
    const long long x = (d + xmy) / 2, y = (d - xmy) / 2;
    const int bx = base_of(c.cx, x - 1), by = base_of(c.cy, y - 1);
    const double *t = c.t;
#pragma unroll
    for (int s = 0; s < S5; s++) o[s] = CP_NEG_INF;
    const double *lower = fcell5_sweep(c, d - 1, xmy - 1);
    const double *middle = fcell5_sweep(c, d - 2, xmy);
    const double *upper = fcell5_sweep(c, d - 1, xmy + 1);
    if (lower) {
        const double eP = e_gap(c.gx, bx);
        o[1] = cp_logAdd(o[1], lower[0] + (eP + t[T5_GAP_SHORT_OPEN_X]));
        o[1] = cp_logAdd(o[1], lower[1] + (eP + t[T5_GAP_SHORT_EXTEND_X]));
        o[3] = cp_logAdd(o[3], lower[0] + (eP + t[T5_GAP_LONG_OPEN_X]));
        o[3] = cp_logAdd(o[3], lower[3] + (eP + t[T5_GAP_LONG_EXTEND_X]));
    }
    if (middle) o[0] = match_from(middle, e_match(c.mm, bx, by), t);
    if (upper) {
        const double eP = e_gap(c.gy, by);
        o[2] = cp_logAdd(o[2], upper[0] + (eP + t[T5_GAP_SHORT_OPEN_Y]));
        o[2] = cp_logAdd(o[2], upper[2] + (eP + t[T5_GAP_SHORT_EXTEND_Y]));
        o[4] = cp_logAdd(o[4], upper[0] + (eP + t[T5_GAP_LONG_OPEN_Y]));
        o[4] = cp_logAdd(o[4], upper[4] + (eP + t[T5_GAP_LONG_EXTEND_Y]));
    }
}

/* gather form of cell_calculateBackward (:378-389) */
__device__ __forceinline__ void backward_cell5(const Ctx5 &c, long long d, long long dTop, int xmy,
                                               double o[S5]) {
    const long long x = (d + xmy) / 2, y = (d - xmy) / 2;
    const double *t = c.t;
#pragma unroll
    for (int s = 0; s < S5; s++) o[s] = CP_NEG_INF;
    /* (ii) cell (x+1, y+1) on d+2 reaches this cell through its middle block */
    const double *s2 = bcell5(c, d + 2, dTop, xmy);
    if (s2) {
        const double eP = e_match(c.mm, base_of(c.cx, x), base_of(c.cy, y));
        o[0] = cp_logAdd(o[0], s2[0] + (eP + t[T5_MATCH_CONTINUE]));
        o[1] = cp_logAdd(o[1], s2[0] + (eP + t[T5_MATCH_FROM_SHORT_GAP_X]));
        o[2] = cp_logAdd(o[2], s2[0] + (eP + t[T5_MATCH_FROM_SHORT_GAP_Y]));
        o[3] = cp_logAdd(o[3], s2[0] + (eP + t[T5_MATCH_FROM_LONG_GAP_X]));
        o[4] = cp_logAdd(o[4], s2[0] + (eP + t[T5_MATCH_FROM_LONG_GAP_Y]));
    }
    /* (iii) cell (x, y+1) on d+1 reaches it through its upper block */
    const double *su = bcell5(c, d + 1, dTop, xmy - 1);
    if (su) {
        const double eP = e_gap(c.gy, base_of(c.cy, y));
        o[0] = cp_logAdd(o[0], su[2] + (eP + t[T5_GAP_SHORT_OPEN_Y]));
        o[2] = cp_logAdd(o[2], su[2] + (eP + t[T5_GAP_SHORT_EXTEND_Y]));
        o[0] = cp_logAdd(o[0], su[4] + (eP + t[T5_GAP_LONG_OPEN_Y]));
        o[4] = cp_logAdd(o[4], su[4] + (eP + t[T5_GAP_LONG_EXTEND_Y]));
    }
    /* (iv) cell (x+1, y) on d+1 reaches it through its lower block */
    const double *sl = bcell5(c, d + 1, dTop, xmy + 1);
    if (sl) {
        const double eP = e_gap(c.gx, base_of(c.cx, x));
        o[0] = cp_logAdd(o[0], sl[1] + (eP + t[T5_GAP_SHORT_OPEN_X]));
        o[1] = cp_logAdd(o[1], sl[1] + (eP + t[T5_GAP_SHORT_EXTEND_X]));
        o[0] = cp_logAdd(o[0], sl[3] + (eP + t[T5_GAP_LONG_OPEN_X]));
        o[3] = cp_logAdd(o[3], sl[3] + (eP + t[T5_GAP_LONG_EXTEND_X]));
    }
}

} // namespace

/* model block: [17 transitions | pad to 24 | 16 match | 4 gapX | 4 gapY] = CP_MODEL5_STRIDE doubles */
extern "C" __global__ __launch_bounds__(256) void cpecan_k_general5(
    const DevItem *items, DevParams P, const int *bandL, const int *bandR,
    const long long *cellPrefix, const char *xChars, const char *yChars, const double *models,
    double *Fstore, double *Bstore, long long *pairs, double *pairLogp, long long *nPairs,
    long long *totXay, double *totVal, long long *nTot, double *dbgB, double *expect) {
    const DevItem it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Ctx5 c;
    c.L = bandL + it.diagBase;
    c.R = bandR + it.diagBase;
    c.pre = cellPrefix + it.diagBase;
    c.cx = xChars + it.xOff;
    c.cy = yChars + it.yOff;
    const double *model = models + (long long) it.model * CP_MODEL5_STRIDE;
    c.t = model;
    c.mm = model + 24;
    c.gx = model + 40;
    c.gy = model + 44;
    c.F = Fstore + it.cellBase * S5;
    c.Bws = Bstore + it.bwsBase;
    c.maxWidth = it.maxWidth;
    extern __shared__ double ldsDiagonals[]; /* the last three forward diagonals, P.ldsWidth cells each (launch-time size) */
    c.ldsW = P.ldsWidth;
    c.ldsF = P.ldsWidth > 0 ? ldsDiagonals : nullptr;
    const double *t = c.t;

    __shared__ double sTotal;
    /* Baum-Welch sums of this alignment: 25 transitions [from*5+to], 80 emissions [state*16+x*4+y] and the
     * likelihood, one copy per wave (LDS atomics), folded into the model's block of `expect` at the end */
    __shared__ double sExp[4][CP_EXPECT5_LEN + 2];
    for (int i = tid; i < 4 * (CP_EXPECT5_LEN + 2); i += 256) (&sExp[0][0])[i] = 0.0;
    const long long D = it.lX + it.lY;
    long long myPairs = 0, myTot = 0;
    if (D == 0) {
        if (tid == 0) { nPairs[blockIdx.x] = 0; nTot[blockIdx.x] = 0; }
        return;
    }
    /* diagonal 0: stateMachine5_startStateProb / raggedStartStateProb (:743-763) */
    if (tid == 0) {
        c.F[0] = it.raggedL ? CP_NEG_INF : 0.0;
        c.F[1] = CP_NEG_INF;
        c.F[2] = CP_NEG_INF;
        c.F[3] = it.raggedL ? 0.0 : CP_NEG_INF;
        c.F[4] = it.raggedL ? 0.0 : CP_NEG_INF;
        if (c.ldsF)
            for (int s = 0; s < S5; s++) c.ldsF[s] = c.F[s];
    }
    __threadfence_block();
    __syncthreads();

    long long tracedBackTo = 0;
    for (long long d = 1; d <= D; d++) {
        const int l = c.L[d], width = ((c.R[d] - l) >> 1) + 1;
        double *fd = c.F + c.pre[d] * S5;
        double *fl = c.ldsF ? c.ldsF + (d % 3) * (long long) c.ldsW * S5 : nullptr;
        for (int cc = tid; cc < width; cc += 256) {
            double o[S5];
            forward_cell5(c, d, l + 2 * cc, o);
#pragma unroll
            for (int s = 0; s < S5; s++) fd[cc * S5 + s] = o[s];
            if (fl)
#pragma unroll
                for (int s = 0; s < S5; s++) fl[cc * S5 + s] = o[s];
        }
        __threadfence_block();
        __syncthreads();

        const bool atEnd = d == D;
        const bool tb = !P.unbanded && d >= tracedBackTo + P.minDiags && width <= P.expansion * 2 + 1;
        if (!(atEnd || tb)) continue;

        /* ---- traceback window (:921-992) ---- */
        const long long dTop = d;
        const long long tracedBackFrom = dTop - (atEnd ? 0 : P.tbDiags + 1);
        {
            double e[S5]; /* stateMachine5_endStateProb / raggedEndStateProb (:765-789) */
            if (atEnd && it.raggedR) {
                e[0] = t[T5_GAP_LONG_OPEN_X];
                e[1] = t[T5_GAP_LONG_OPEN_X];
                e[2] = t[T5_GAP_LONG_OPEN_Y];
                e[3] = t[T5_GAP_LONG_EXTEND_X];
                e[4] = t[T5_GAP_LONG_EXTEND_Y];
            } else {
                e[0] = t[T5_MATCH_CONTINUE];
                e[1] = t[T5_MATCH_FROM_SHORT_GAP_X];
                e[2] = t[T5_MATCH_FROM_SHORT_GAP_Y];
                e[3] = t[T5_MATCH_FROM_LONG_GAP_X];
                e[4] = t[T5_MATCH_FROM_LONG_GAP_Y];
            }
            double *b = bslot5(c, dTop);
            for (int cc = tid; cc < width; cc += 256)
#pragma unroll
                for (int s = 0; s < S5; s++) b[cc * S5 + s] = e[s];
        }
        __threadfence_block();
        __syncthreads();

        double total = CP_NEG_INF;
        long long calcs = 0;
        for (long long d2 = dTop; d2 > tracedBackTo; d2--) {
            const int l2 = c.L[d2], w2 = ((c.R[d2] - l2) >> 1) + 1;
            if (d2 < dTop) {
                double *b = bslot5(c, d2);
                for (int cc = tid; cc < w2; cc += 256) {
                    double o[S5];
                    backward_cell5(c, d2, dTop, l2 + 2 * cc, o);
#pragma unroll
                    for (int s = 0; s < S5; s++) b[cc * S5 + s] = o[s];
                }
                __threadfence_block();
                __syncthreads();
            }
            if (d2 > tracedBackFrom) continue;

            const double *fdd = c.F + c.pre[d2] * S5;
            const double *bdd = bslot5(c, d2);
            if (P.unbanded ? calcs++ == 0 : calcs++ % 10 == 0) {
                /* diagonalCalculationTotalProbability :736-754, by wave 0 */
                if (wave == 0) {
                    double acc = CP_NEG_INF;
                    for (int base = 0; base < w2; base += 64) {
                        const int cc = base + lane;
                        const bool valid = cc < w2;
                        double v = CP_NEG_INF;
                        if (valid) { /* cell_dotProduct :391-397 */
                            v = fdd[cc * S5] + bdd[cc * S5];
#pragma unroll
                            for (int s = 1; s < S5; s++) v = cp_logAdd(v, fdd[cc * S5 + s] + bdd[cc * S5 + s]);
                        }
                        acc = cp_wave_seq_fold(acc, v, valid);
                    }
                    if (d2 + 1 <= dTop) {
                        /* matches that step over d2: forward[d2-1] --match--> cells of d2+1 */
                        const int l3 = c.L[d2 + 1], w3 = ((c.R[d2 + 1] - l3) >> 1) + 1;
                        const double *b3 = bslot5(c, d2 + 1);
                        double acc2 = CP_NEG_INF;
                        for (int base = 0; base < w3; base += 64) {
                            const int cc = base + lane;
                            const bool valid = cc < w3;
                            double v = CP_NEG_INF;
                            if (valid) {
                                const int xmy = l3 + 2 * cc;
                                const double *mid = fcell5(c, d2 - 1, xmy);
                                double m = CP_NEG_INF;
                                if (mid) {
                                    const long long x = (d2 + 1 + xmy) / 2, y = (d2 + 1 - xmy) / 2;
                                    m = match_from(mid, e_match(c.mm, base_of(c.cx, x - 1), base_of(c.cy, y - 1)), t);
                                }
                                v = m + b3[cc * S5];
#pragma unroll
                                for (int s = 1; s < S5; s++) v = cp_logAdd(v, CP_NEG_INF + b3[cc * S5 + s]);
                            }
                            acc2 = cp_wave_seq_fold(acc2, v, valid);
                        }
                        acc = cp_logAdd(acc, acc2);
                    }
                    if (lane == 0) {
                        sTotal = acc;
                        if (myTot < it.totCap) {
                            totXay[it.totBase + myTot] = d2;
                            totVal[it.totBase + myTot] = acc;
                        }
                    }
                    myTot++;
                }
                __syncthreads();
                total = sTotal;
                __syncthreads();
            }

            if (P.debug && dbgB) {
                double *o = dbgB + (it.cellBase + c.pre[d2]) * S5;
                for (int cc = tid; cc < w2 * S5; cc += 256) o[cc] = bdd[cc];
            }

            if (P.mode == 1) {
                /* diagonalCalculation_Expectations :841-863 over stateMachine5_cellCalculate with
                 * cell_updateExpectations (:407-424): every transition into a cell of backward[d2] from its
                 * forward neighbours adds p = exp(from + to + (eP + tP) - total) to its transition count and,
                 * unless a base is not ACGT, to the emission count [to][x][y] */
                double *acc = sExp[wave];
                if (tid == 0) acc[CP_EXPECT5_LEN - 1] += total; /* likelihood, once per diagonal (quirk Q7) */
                const bool haveMiddle = d2 - 2 >= tracedBackTo; /* forward[d2-2] is freed otherwise (:982) */
                for (int cc = tid; cc < w2; cc += 256) {
                    const int xmy = l2 + 2 * cc;
                    const long long x = (d2 + xmy) / 2, y = (d2 - xmy) / 2;
                    const int bx = base_of(c.cx, x - 1), by = base_of(c.cy, y - 1);
                    const double *cur = bdd + cc * S5;
                    const double *lower = fcell5(c, d2 - 1, xmy - 1);
                    const double *middle = haveMiddle ? fcell5(c, d2 - 2, xmy) : nullptr;
                    const double *upper = fcell5(c, d2 - 1, xmy + 1);
                    double into[S5] = { 0.0, 0.0, 0.0, 0.0, 0.0 }; /* per to-state sums for the emission counts */
                    auto tr = [&](const double *nb, int f, int to, double eP, int ti) {
                        const double pr = exp(nb[f] + cur[to] + (eP + t[ti]) - total);
                        atomicAdd(&acc[f * S5 + to], pr);
                        into[to] += pr;
                    };
                    if (lower) {
                        const double eP = e_gap(c.gx, bx);
                        tr(lower, 0, 1, eP, T5_GAP_SHORT_OPEN_X);
                        tr(lower, 1, 1, eP, T5_GAP_SHORT_EXTEND_X);
                        tr(lower, 0, 3, eP, T5_GAP_LONG_OPEN_X);
                        tr(lower, 3, 3, eP, T5_GAP_LONG_EXTEND_X);
                    }
                    if (middle) {
                        const double eP = e_match(c.mm, bx, by);
                        tr(middle, 0, 0, eP, T5_MATCH_CONTINUE);
                        tr(middle, 1, 0, eP, T5_MATCH_FROM_SHORT_GAP_X);
                        tr(middle, 2, 0, eP, T5_MATCH_FROM_SHORT_GAP_Y);
                        tr(middle, 3, 0, eP, T5_MATCH_FROM_LONG_GAP_X);
                        tr(middle, 4, 0, eP, T5_MATCH_FROM_LONG_GAP_Y);
                    }
                    if (upper) {
                        const double eP = e_gap(c.gy, by);
                        tr(upper, 0, 2, eP, T5_GAP_SHORT_OPEN_Y);
                        tr(upper, 2, 2, eP, T5_GAP_SHORT_EXTEND_Y);
                        tr(upper, 0, 4, eP, T5_GAP_LONG_OPEN_Y);
                        tr(upper, 4, 4, eP, T5_GAP_LONG_EXTEND_Y);
                    }
                    if (bx < 4 && by < 4) {
#pragma unroll
                        for (int st = 0; st < S5; st++)
                            if (into[st] != 0.0) atomicAdd(&acc[25 + st * 16 + bx * 4 + by], into[st]);
                    }
                }
                __syncthreads();
                continue;
            }

            /* diagonalCalculationPosteriorMatchProbs :756-795, ordered emission by wave 0 */
            if (wave == 0) {
                for (int base = 0; base < w2; base += 64) {
                    const int cc = base + lane;
                    bool hit = false;
                    double e = 0.0, p = 0.0;
                    long long x = 0, y = 0;
                    if (cc < w2) {
                        const int xmy = l2 + 2 * cc;
                        x = (d2 + xmy) / 2;
                        y = (d2 - xmy) / 2;
                        if (x > 0 && y > 0) {
                            e = (fdd[cc * S5] + bdd[cc * S5]) - total;
                            p = exp(e);
                            hit = p >= P.threshold;
                        }
                    }
                    const unsigned long long m = __ballot(hit);
                    if (hit) {
                        const long long idx = myPairs + __popcll(m & ((1ull << lane) - 1ull));
                        if (idx < it.pairCap) {
                            if (p > 1.0) p = 1.0;
                            long long *o = pairs + (it.pairBase + idx) * 3;
                            o[0] = (long long) floor(p * 10000000.0);
                            o[1] = x - 1;
                            o[2] = y - 1;
                            pairLogp[it.pairBase + idx] = e;
                        }
                    }
                    myPairs += __popcll(m);
                }
            }
            __syncthreads();
        }
        tracedBackTo = tracedBackFrom;
    }
    if (P.mode == 1 && expect) {
        __syncthreads();
        double *dst = expect + (long long) it.model * CP_EXPECT5_LEN;
        for (int i = tid; i < CP_EXPECT5_LEN; i += 256) {
            const double v = ((sExp[0][i] + sExp[1][i]) + sExp[2][i]) + sExp[3][i];
            if (v != 0.0) atomicAdd(dst + i, v);
        }
    }
    if (tid == 0) {
        nPairs[blockIdx.x] = myPairs;
        nTot[blockIdx.x] = myTot;
    }
}

/*
 * cpecan_kernel_general4.hip -- banded forward / backward / posterior DP for the reference's 4-state signal machine
 * (stateMachine4_cellCalculate, impl/stateMachine.c:867-897; getStateMachine4 :1750-1759; vanillaAlign.c:122-123):
 * match, short gap X, short gap Y and a long gap X over the strawMan emissions -- the k-mer / event Gaussian pair for
 * a match and, with the extra-event table, for a gap in Y; the k-mer gap table for a gap in X.
 *
 * Structure of cpecan_kernel_general.hip: one 256-thread workgroup per work item (= one getPosteriorProbsWithBanding
 * call, impl/pairwiseAligner.c:870-1006), any band width, forward diagonals in HBM as [cell][state], three rotating
 * backward diagonals; the backward recurrence is the gather form of the reference's scatter, contributions added in the
 * reference's order per target state.  Posterior decode only: the reference has no Hmm container for this machine
 * (hmmContinuous_getEmptyHmm, impl/continuousHmm.c:913-945, knows threeState, threeStateHdp and vanilla).
 *
 * The model is a strawMan table (cpecan_hip.hip: derive_rows) whose header holds the machine's eleven transitions in the
 * member order of _StateMachine4 (inc/stateMachine.h:134-152).
 */
#include "cpecan_device.h"

namespace {

enum { /* _StateMachine4's members in order */
    T4_MATCH_CONTINUE = 0, T4_MATCH_FROM_SHORT_GAP_X, T4_MATCH_FROM_LONG_GAP_X, T4_MATCH_FROM_SHORT_GAP_Y,
    T4_GAP_SHORT_OPEN_X, T4_GAP_SHORT_EXTEND_X, T4_GAP_SHORT_OPEN_Y, T4_GAP_SHORT_EXTEND_Y,
    T4_GAP_LONG_OPEN_X, T4_GAP_LONG_EXTEND_X, T4_GAP_LONG_SWITCH_TO_X
};
/* states (inc/stateMachine.h:31-33): match 0, shortGapX 1, shortGapY 2, longGapX 3 */
constexpr int S4 = 4;

struct Ctx4 {
    const int *L, *R;
    const long long *pre;
    const unsigned short *kidx;
    const double *ev;
    const double *rows;
    const double *t;
    double *F, *Bws;
    int maxWidth;
};

__device__ __forceinline__ const double *row4(const Ctx4 &c, long long ix) {
    const int k = ix >= 0 ? (int) c.kidx[ix] : 4096; /* index < 0: the "n" sentinel (impl/pairwiseAligner.c:314-318) */
    return c.rows + (long long) k * CP_ROW;
}
__device__ __forceinline__ void event4(const Ctx4 &c, long long iy, double &mean, double &noise) {
    if (iy >= 0) { mean = c.ev[3 * iy]; noise = c.ev[3 * iy + 1]; }
    else { mean = CP_NEG_INF; noise = 0.0; } /* NULLEVENT (:261) */
}
/* emissions_signal_strawManGetKmerEventMatchProb (impl/stateMachine.c:595-629) over the match / extra-event table */
__device__ __forceinline__ double emitM4(const double *r, double mean, double noise) {
    return cp_logGauss(mean, r[CP_MU], r[CP_SD], r[CP_K1]) + cp_logGauss(noise, r[CP_NMU], r[CP_NSD], r[CP_K2]);
}
__device__ __forceinline__ double emitY4(const double *r, double mean, double noise) {
    return cp_logGauss(mean, r[CP_YMU], r[CP_YSD], r[CP_YK1]) + cp_logGauss(noise, r[CP_YNMU], r[CP_YNSD], r[CP_YK2]);
}
__device__ __forceinline__ const double *fcell4(const Ctx4 &c, long long d, int xmy) {
    if (d < 0) return nullptr;
    const int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    return c.F + (c.pre[d] + ((xmy - l) >> 1)) * S4;
}
__device__ __forceinline__ double *bslot4(const Ctx4 &c, long long d) { return c.Bws + (d % 3) * (long long) c.maxWidth * S4; }
__device__ __forceinline__ const double *bcell4(const Ctx4 &c, long long d, long long dTop, int xmy) {
    if (d > dTop) return nullptr;
    const int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    return bslot4(c, d) + ((xmy - l) >> 1) * S4;
}

/* the match state's incoming sum from the cell (x-1, y-1): the middle block of stateMachine4_cellCalculate (:884-890) */
__device__ __forceinline__ double match_from4(const double *middle, double eP, const double *t) {
    double m = CP_NEG_INF;
    m = cp_logAdd(m, middle[0] + (eP + t[T4_MATCH_CONTINUE]));
    m = cp_logAdd(m, middle[1] + (eP + t[T4_MATCH_FROM_SHORT_GAP_X]));
    m = cp_logAdd(m, middle[2] + (eP + t[T4_MATCH_FROM_SHORT_GAP_Y]));
    m = cp_logAdd(m, middle[3] + (eP + t[T4_MATCH_FROM_LONG_GAP_X]));
    return m;
}

/* cell_calculateForward (impl/pairwiseAligner.c:365-375) over stateMachine4_cellCalculate */
__device__ __forceinline__ void forward_cell4(const Ctx4 &c, long long d, int xmy, double o[S4]) {
    const long long x = (d + xmy) / 2, y = (d - xmy) / 2;
    const double *row = row4(c, x - 1);
    double mean, noise;
    event4(c, y - 1, mean, noise);
    const double *t = c.t;
    double m = CP_NEG_INF, sx = CP_NEG_INF, sy = CP_NEG_INF, lx = CP_NEG_INF;
    const double *lower = fcell4(c, d - 1, xmy - 1), *middle = fcell4(c, d - 2, xmy), *upper = fcell4(c, d - 1, xmy + 1);
    if (lower) {
        const double eP = row[CP_GAPX];
        sx = cp_logAdd(sx, lower[0] + (eP + t[T4_GAP_SHORT_OPEN_X]));
        sx = cp_logAdd(sx, lower[1] + (eP + t[T4_GAP_SHORT_EXTEND_X]));
        lx = cp_logAdd(lx, lower[0] + (eP + t[T4_GAP_LONG_OPEN_X]));
        lx = cp_logAdd(lx, lower[3] + (eP + t[T4_GAP_LONG_EXTEND_X]));
        lx = cp_logAdd(lx, lower[2] + (eP + t[T4_GAP_LONG_SWITCH_TO_X]));
    }
    if (middle) m = match_from4(middle, emitM4(row, mean, noise), t);
    if (upper) {
        const double eP = emitY4(row, mean, noise);
        sy = cp_logAdd(sy, upper[0] + (eP + t[T4_GAP_SHORT_OPEN_Y]));
        sy = cp_logAdd(sy, upper[2] + (eP + t[T4_GAP_SHORT_EXTEND_Y]));
    }
    o[0] = m; o[1] = sx; o[2] = sy; o[3] = lx;
}

/* gather form of cell_calculateBackward (:378-389): what reaches this cell from the cell above-right on d+2 (its middle
 * block), from (x, y+1) on d+1 (its upper block) and from (x+1, y) on d+1 (its lower block), in that order; within a
 * block in the order of stateMachine4_cellCalculate's calls */
__device__ __forceinline__ void backward_cell4(const Ctx4 &c, long long d, long long dTop, int xmy, double o[S4]) {
    const long long x = (d + xmy) / 2, y = (d - xmy) / 2;
    const double *t = c.t;
    double m = CP_NEG_INF, sx = CP_NEG_INF, sy = CP_NEG_INF, lx = CP_NEG_INF;
    const double *s2 = bcell4(c, d + 2, dTop, xmy);
    if (s2) {
        double mean, noise;
        event4(c, y, mean, noise);
        const double eP = emitM4(row4(c, x), mean, noise);
        m = cp_logAdd(m, s2[0] + (eP + t[T4_MATCH_CONTINUE]));
        sx = cp_logAdd(sx, s2[0] + (eP + t[T4_MATCH_FROM_SHORT_GAP_X]));
        sy = cp_logAdd(sy, s2[0] + (eP + t[T4_MATCH_FROM_SHORT_GAP_Y]));
        lx = cp_logAdd(lx, s2[0] + (eP + t[T4_MATCH_FROM_LONG_GAP_X]));
    }
    const double *su = bcell4(c, d + 1, dTop, xmy - 1);
    if (su) {
        double mean, noise;
        event4(c, y, mean, noise);
        const double eP = emitY4(row4(c, x - 1), mean, noise);
        m = cp_logAdd(m, su[2] + (eP + t[T4_GAP_SHORT_OPEN_Y]));
        sy = cp_logAdd(sy, su[2] + (eP + t[T4_GAP_SHORT_EXTEND_Y]));
    }
    const double *sl = bcell4(c, d + 1, dTop, xmy + 1);
    if (sl) {
        const double eP = row4(c, x)[CP_GAPX];
        m = cp_logAdd(m, sl[1] + (eP + t[T4_GAP_SHORT_OPEN_X]));
        sx = cp_logAdd(sx, sl[1] + (eP + t[T4_GAP_SHORT_EXTEND_X]));
        m = cp_logAdd(m, sl[3] + (eP + t[T4_GAP_LONG_OPEN_X]));
        lx = cp_logAdd(lx, sl[3] + (eP + t[T4_GAP_LONG_EXTEND_X]));
        sy = cp_logAdd(sy, sl[3] + (eP + t[T4_GAP_LONG_SWITCH_TO_X]));
    }
    o[0] = m; o[1] = sx; o[2] = sy; o[3] = lx;
}

} // namespace

extern "C" __global__ __launch_bounds__(256) void cpecan_k_general4(
    const DevItem *items, DevParams P, const int *bandL, const int *bandR, const long long *cellPrefix,
    const unsigned short *kidx, const double *events, const double *models, double *Fstore, double *Bstore,
    long long *pairs, double *pairLogp, long long *nPairs, long long *totXay, double *totVal, long long *nTot) {
    const DevItem it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Ctx4 c;
    c.L = bandL + it.diagBase;
    c.R = bandR + it.diagBase;
    c.pre = cellPrefix + it.diagBase;
    c.kidx = kidx + it.xOff;
    c.ev = events + 3 * it.yOff;
    const double *model = models + (long long) it.model * CP_MODEL_STRIDE;
    c.t = model;
    c.rows = model + CP_MODEL_HEADER;
    c.F = Fstore + it.cellBase * S4;
    c.Bws = Bstore + it.bwsBase;
    c.maxWidth = it.maxWidth;
    __shared__ double sTotal;

    const long long D = it.lX + it.lY;
    long long myPairs = 0, myTot = 0;
    if (D == 0) {
        if (tid == 0) { nPairs[blockIdx.x] = 0; nTot[blockIdx.x] = 0; }
        return;
    }
    const double *t = c.t;
    if (tid == 0) { /* stateMachine5_startStateProb / stateMachine4_raggedStartStateProb (:743, :791) */
        c.F[0] = it.raggedL ? CP_NEG_INF : 0.0;
        c.F[1] = CP_NEG_INF;
        c.F[2] = it.raggedL ? 0.0 : CP_NEG_INF;
        c.F[3] = it.raggedL ? 0.0 : CP_NEG_INF;
    }
    __threadfence_block();
    __syncthreads();

    long long tracedBackTo = 0;
    for (long long d = 1; d <= D; d++) {
        const int l = c.L[d], width = ((c.R[d] - l) >> 1) + 1;
        double *fd = c.F + c.pre[d] * S4;
        for (int cc = tid; cc < width; cc += 256) {
            double o[S4];
            forward_cell4(c, d, l + 2 * cc, o);
            for (int s = 0; s < S4; s++) fd[cc * S4 + s] = o[s];
        }
        __threadfence_block();
        __syncthreads();
        const bool atEnd = d == D;
        const bool tb = !P.unbanded && d >= tracedBackTo + P.minDiags && width <= P.expansion * 2 + 1;
        if (!(atEnd || tb)) continue;

        const long long dTop = d, tracedBackFrom = dTop - (atEnd ? 0 : P.tbDiags + 1);
        {
            double e[S4]; /* stateMachine4_endStateProb / _raggedEndStateProb (:796-829) */
            if (atEnd && it.raggedR) {
                e[0] = e[1] = e[2] = t[T4_GAP_LONG_OPEN_X];
                e[3] = t[T4_GAP_LONG_EXTEND_X];
            } else {
                e[0] = t[T4_MATCH_CONTINUE];
                e[1] = t[T4_MATCH_FROM_SHORT_GAP_X];
                e[2] = t[T4_MATCH_FROM_SHORT_GAP_Y];
                e[3] = t[T4_MATCH_FROM_LONG_GAP_X];
            }
            double *b = bslot4(c, dTop);
            for (int cc = tid; cc < width; cc += 256)
                for (int s = 0; s < S4; s++) b[cc * S4 + s] = e[s];
        }
        __threadfence_block();
        __syncthreads();

        double total = CP_NEG_INF;
        long long calcs = 0;
        for (long long d2 = dTop; d2 > tracedBackTo; d2--) {
            const int l2 = c.L[d2], w2 = ((c.R[d2] - l2) >> 1) + 1;
            if (d2 < dTop) {
                double *b = bslot4(c, d2);
                for (int cc = tid; cc < w2; cc += 256) {
                    double o[S4];
                    backward_cell4(c, d2, dTop, l2 + 2 * cc, o);
                    for (int s = 0; s < S4; s++) b[cc * S4 + s] = o[s];
                }
                __threadfence_block();
                __syncthreads();
            }
            if (d2 > tracedBackFrom) continue;
            const double *fdd = c.F + c.pre[d2] * S4, *bdd = bslot4(c, d2);
            if (P.unbanded ? calcs++ == 0 : calcs++ % 10 == 0) {
                if (wave == 0) { /* diagonalCalculationTotalProbability :736-754 */
                    double acc = CP_NEG_INF;
                    for (int base = 0; base < w2; base += 64) {
                        const int cc = base + lane;
                        const bool valid = cc < w2;
                        double v = CP_NEG_INF;
                        if (valid) { /* cell_dotProduct :391-397 */
                            v = fdd[cc * S4] + bdd[cc * S4];
                            for (int s = 1; s < S4; s++) v = cp_logAdd(v, fdd[cc * S4 + s] + bdd[cc * S4 + s]);
                        }
                        acc = cp_wave_seq_fold(acc, v, valid);
                    }
                    if (d2 + 1 <= dTop) { /* matches that step over d2 */
                        const int l3 = c.L[d2 + 1], w3 = ((c.R[d2 + 1] - l3) >> 1) + 1;
                        const double *b3 = bslot4(c, d2 + 1);
                        double acc2 = CP_NEG_INF;
                        for (int base = 0; base < w3; base += 64) {
                            const int cc = base + lane;
                            const bool valid = cc < w3;
                            double v = CP_NEG_INF;
                            if (valid) {
                                const int xmy = l3 + 2 * cc;
                                const double *mid = fcell4(c, d2 - 1, xmy);
                                double mm = CP_NEG_INF;
                                if (mid) {
                                    const long long x = (d2 + 1 + xmy) / 2, y = (d2 + 1 - xmy) / 2;
                                    double mean, noise;
                                    event4(c, y - 1, mean, noise);
                                    mm = match_from4(mid, emitM4(row4(c, x - 1), mean, noise), t);
                                }
                                v = mm + b3[cc * S4];
                                for (int s = 1; s < S4; s++) v = cp_logAdd(v, CP_NEG_INF + b3[cc * S4 + s]);
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
            if (wave == 0) { /* diagonalCalculationPosteriorMatchProbs :756-795, in order */
                for (int base = 0; base < w2; base += 64) {
                    const int cc = base + lane;
                    bool hit = false;
                    double e = 0.0, p = 0.0;
                    long long x = 0, y = 0;
                    if (cc < w2) {
                        const int xmy = l2 + 2 * cc;
                        x = (d2 + xmy) / 2; y = (d2 - xmy) / 2;
                        if (x > 0 && y > 0) {
                            e = (fdd[cc * S4] + bdd[cc * S4]) - total;
                            p = exp(e);
                            hit = p >= P.threshold;
                        }
                    }
                    const unsigned long long mk = __ballot(hit);
                    if (hit) {
                        const long long idx = myPairs + __popcll(mk & ((1ull << lane) - 1ull));
                        if (idx < it.pairCap) {
                            if (p > 1.0) p = 1.0;
                            long long *o = pairs + (it.pairBase + idx) * 3;
                            o[0] = (long long) floor(p * 10000000.0);
                            o[1] = x - 1;
                            o[2] = y - 1;
                            pairLogp[it.pairBase + idx] = e;
                        }
                    }
                    myPairs += __popcll(mk);
                }
            }
            __syncthreads();
        }
        tracedBackTo = tracedBackFrom;
    }
    if (tid == 0) {
        nPairs[blockIdx.x] = myPairs;
        nTot[blockIdx.x] = myTot;
    }
}

/*
 * cpecan_kernel_general.hip -- banded forward / backward / posterior DP for bands of ANY width.
 *
 * One 256-thread workgroup per work item (= one getPosteriorProbsWithBanding call,
 * impl/pairwiseAligner.c:870-1006).  Threads stride over the cells of the current anti-diagonal;
 * forward diagonals are stored in HBM ([cell][state], the reference's DpDiagonal layout :567), the
 * three live backward diagonals rotate through a small HBM workspace (L2-resident).  This is the
 * correctness-first kernel and the fall-back for bands wider than the systolic kernel's 192 k-mers
 * (un-banded alignments, sparse anchors); the batch throughput path is cpecan_kernel_systolic.hip.
 *
 * Differences from the reference's control flow, none of which changes a result bit:
 *   - backward is a gather (the reference scatters, :378-389); per target cell the contributions
 *     are added in the reference's order: from (d+2, xmy) [middle block], then from (d+1, xmy-1)
 *     [its upper block], then from (d+1, xmy+1) [its lower block];
 *   - a neighbour outside the band contributes -inf instead of being skipped (logAdd(a,-inf)==a);
 *   - totalProbability's sequential fold visits only the terms that change the running value
 *     (cp_wave_seq_fold).
 */
#include "cpecan_device.h"

namespace {

struct Ctx {
    const int *L, *R;            /* band of this item */
    const long long *pre;        /* cell prefix per diagonal */
    const unsigned short *kidx;  /* k-mer index per X element (0..4095, 4096 = invalid) */
    const double *ev;            /* events of this item, 3 doubles each */
    const double *rows;          /* model rows */
    const double *t;             /* 9 transitions */
    double *F;                   /* forward cells of this item */
    double *Bws;                 /* 3 x maxWidth x 3 backward workspace */
    long long lX, lY;
    int maxWidth;
};

__device__ __forceinline__ const double *row_of(const Ctx &c, long long ix) {
    /* X element ix-1 ... caller passes the sequence index; index < 0 is the "n" sentinel (:314-318) */
    int k = ix >= 0 ? (int) c.kidx[ix] : 4096;
    return c.rows + (long long) k * CP_ROW;
}

__device__ __forceinline__ void event_of(const Ctx &c, long long iy, double &mean, double &noise) {
    /* index < 0 is NULLEVENT = {-inf, 0} (:261,:333-337) */
    if (iy >= 0) {
        mean = c.ev[3 * iy];
        noise = c.ev[3 * iy + 1];
    } else {
        mean = CP_NEG_INF;
        noise = 0.0;
    }
}

/* emissions_signal_strawManGetKmerEventMatchProb impl/stateMachine.c:595-629 */
__device__ __forceinline__ double emit_match(const double *r, double mean, double noise) {
    double a = cp_logGauss(mean, r[CP_MU], r[CP_SD], r[CP_K1]);
    double b = cp_logGauss(noise, r[CP_NMU], r[CP_NSD], r[CP_K2]);
    return a + b;
}
__device__ __forceinline__ double emit_gapy(const double *r, double mean, double noise) {
    double a = cp_logGauss(mean, r[CP_YMU], r[CP_YSD], r[CP_YK1]);
    double b = cp_logGauss(noise, r[CP_YNMU], r[CP_YNSD], r[CP_YK2]);
    return a + b;
}

__device__ __forceinline__ const double *fcell(const Ctx &c, long long d, int xmy) {
    if (d < 0) return nullptr;
    int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    return c.F + (c.pre[d] + ((xmy - l) >> 1)) * 3;
}
__device__ __forceinline__ double *bslot(const Ctx &c, long long d) {
    return c.Bws + (d % 3) * (long long) c.maxWidth * 3;
}
__device__ __forceinline__ const double *bcell(const Ctx &c, long long d, long long dTop, int xmy) {
    if (d > dTop) return nullptr;
    int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    return bslot(c, d) + ((xmy - l) >> 1) * 3;
}

/* forward cell: cell_calculateForward + stateMachine3_cellCalculate (impl/stateMachine.c:1305-1334) */
__device__ __forceinline__ void forward_cell(const Ctx &c, long long d, int xmy, double out[3]) {
    long long x = (d + xmy) / 2, y = (d - xmy) / 2;
    const double *row = row_of(c, x - 1);
    double mean, noise;
    event_of(c, y - 1, mean, noise);
    const double *t = c.t;
    double m = CP_NEG_INF, gx = CP_NEG_INF, gy = CP_NEG_INF;
    const double *lower = fcell(c, d - 1, xmy - 1);
    const double *middle = fcell(c, d - 2, xmy);
    const double *upper = fcell(c, d - 1, xmy + 1);
    if (lower) {
        double eP = row[CP_GAPX];
        gx = cp_logAdd(gx, lower[0] + (eP + t[T_GAP_OPEN_X]));
        gx = cp_logAdd(gx, lower[1] + (eP + t[T_GAP_EXTEND_X]));
        gx = cp_logAdd(gx, lower[2] + (eP + t[T_GAP_SWITCH_TO_X]));
    }
    if (middle) {
        double eP = emit_match(row, mean, noise);
        m = cp_logAdd(m, middle[0] + (eP + t[T_MATCH_CONTINUE]));
        m = cp_logAdd(m, middle[1] + (eP + t[T_MATCH_FROM_GAP_X]));
        m = cp_logAdd(m, middle[2] + (eP + t[T_MATCH_FROM_GAP_Y]));
    }
    if (upper) {
        double eP = emit_gapy(row, mean, noise);
        gy = cp_logAdd(gy, upper[0] + (eP + t[T_GAP_OPEN_Y]));
        gy = cp_logAdd(gy, upper[2] + (eP + t[T_GAP_EXTEND_Y]));
    }
    out[0] = m; out[1] = gx; out[2] = gy;
}

/* backward cell, gather form of cell_calculateBackward (:378-389) */
__device__ __forceinline__ void backward_cell(const Ctx &c, long long d, long long dTop, int xmy,
                                              double out[3]) {
    long long x = (d + xmy) / 2, y = (d - xmy) / 2;
    const double *t = c.t;
    double m = CP_NEG_INF, gx = CP_NEG_INF, gy = CP_NEG_INF;
    /* (ii) cell (x+1,y+1) on d+2 reaches this cell through its middle block */
    const double *s2 = bcell(c, d + 2, dTop, xmy);
    if (s2) {
        const double *row = row_of(c, x);
        double mean, noise;
        event_of(c, y, mean, noise);
        double eP = emit_match(row, mean, noise);
        m = cp_logAdd(m, s2[0] + (eP + t[T_MATCH_CONTINUE]));
        gx = cp_logAdd(gx, s2[0] + (eP + t[T_MATCH_FROM_GAP_X]));
        gy = cp_logAdd(gy, s2[0] + (eP + t[T_MATCH_FROM_GAP_Y]));
    }
    /* (iii) cell (x,y+1) on d+1 reaches it through its upper block */
    const double *su = bcell(c, d + 1, dTop, xmy - 1);
    if (su) {
        const double *row = row_of(c, x - 1);
        double mean, noise;
        event_of(c, y, mean, noise);
        double eP = emit_gapy(row, mean, noise);
        m = cp_logAdd(m, su[2] + (eP + t[T_GAP_OPEN_Y]));
        gy = cp_logAdd(gy, su[2] + (eP + t[T_GAP_EXTEND_Y]));
    }
    /* (iv) cell (x+1,y) on d+1 reaches it through its lower block */
    const double *sl = bcell(c, d + 1, dTop, xmy + 1);
    if (sl) {
        const double *row = row_of(c, x);
        double eP = row[CP_GAPX];
        m = cp_logAdd(m, sl[1] + (eP + t[T_GAP_OPEN_X]));
        gx = cp_logAdd(gx, sl[1] + (eP + t[T_GAP_EXTEND_X]));
        gy = cp_logAdd(gy, sl[1] + (eP + t[T_GAP_SWITCH_TO_X]));
    }
    out[0] = m; out[1] = gx; out[2] = gy;
}

} // namespace

extern "C" __global__ __launch_bounds__(256) void cpecan_k_general(
    const DevItem *items, DevParams P, const int *bandL, const int *bandR,
    const long long *cellPrefix, const unsigned short *kidx, const double *events,
    const double *models, double *Fstore, double *Bstore, long long *pairs, double *pairLogp,
    long long *nPairs, long long *totXay, double *totVal, long long *nTot, double *dbgB,
    double *expect) {
    const DevItem it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Ctx c;
    c.L = bandL + it.diagBase;
    c.R = bandR + it.diagBase;
    c.pre = cellPrefix + it.diagBase;
    c.kidx = kidx + it.xOff;
    c.ev = events + 3 * it.yOff;
    const double *model = models + (long long) it.model * CP_MODEL_STRIDE;
    c.t = model;
    c.rows = model + CP_MODEL_HEADER;
    c.F = Fstore + it.cellBase * 3;
    c.Bws = Bstore + it.bwsBase;
    c.lX = it.lX;
    c.lY = it.lY;
    c.maxWidth = it.maxWidth;

    __shared__ double sTotal;
    __shared__ double sExp[16];

    const long long D = it.lX + it.lY;
    long long myPairs = 0, myTot = 0; /* wave-0 uniform counters */
    if (D == 0) {
        if (tid == 0) { nPairs[blockIdx.x] = 0; nTot[blockIdx.x] = 0; }
        return;
    }
    const double *t = c.t;

    /* diagonal 0: start state vector (stateMachine3_startStateProb / raggedStart :1168-1177) */
    if (tid == 0) {
        c.F[0] = it.raggedL ? CP_NEG_INF : 0.0;
        c.F[1] = it.raggedL ? 0.0 : CP_NEG_INF;
        c.F[2] = it.raggedL ? 0.0 : CP_NEG_INF;
    }
    __threadfence_block();
    __syncthreads();

    double expAcc[10]; /* per-thread partial expectations: 9 transitions + likelihood */
    for (int i = 0; i < 10; i++) expAcc[i] = 0.0;
    double *gapAcc = expect ? expect + (long long) it.model * (9 + 4096 + 1) + 9 : nullptr;

    long long tracedBackTo = 0;
    for (long long d = 1; d <= D; d++) {
        const int l = c.L[d], width = ((c.R[d] - l) >> 1) + 1;
        double *fd = c.F + c.pre[d] * 3;
        for (int cc = tid; cc < width; cc += 256) {
            double o[3];
            forward_cell(c, d, l + 2 * cc, o);
            fd[cc * 3] = o[0]; fd[cc * 3 + 1] = o[1]; fd[cc * 3 + 2] = o[2];
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
            double e0, e1, e2; /* end state vector :1179-1207 */
            if (atEnd && it.raggedR) {
                e0 = (t[T_GAP_OPEN_X] + t[T_GAP_OPEN_Y]) / 2.0;
                e1 = t[T_GAP_EXTEND_X];
                e2 = t[T_GAP_EXTEND_Y];
            } else {
                e0 = t[T_MATCH_CONTINUE];
                e1 = t[T_MATCH_FROM_GAP_X];
                e2 = t[T_MATCH_FROM_GAP_Y];
            }
            double *b = bslot(c, dTop);
            for (int cc = tid; cc < width; cc += 256) {
                b[cc * 3] = e0; b[cc * 3 + 1] = e1; b[cc * 3 + 2] = e2;
            }
        }
        __threadfence_block();
        __syncthreads();

        double total = CP_NEG_INF;
        long long calcs = 0;
        for (long long d2 = dTop; d2 > tracedBackTo; d2--) {
            const int l2 = c.L[d2], w2 = ((c.R[d2] - l2) >> 1) + 1;
            if (d2 < dTop) {
                double *b = bslot(c, d2);
                for (int cc = tid; cc < w2; cc += 256) {
                    double o[3];
                    backward_cell(c, d2, dTop, l2 + 2 * cc, o);
                    b[cc * 3] = o[0]; b[cc * 3 + 1] = o[1]; b[cc * 3 + 2] = o[2];
                }
                __threadfence_block();
                __syncthreads();
            }
            if (d2 > tracedBackFrom) continue;

            const double *fdd = c.F + c.pre[d2] * 3;
            const double *bdd = bslot(c, d2);
            /* banded: refreshed every 10th posterior diagonal of the window (:956); un-banded:
             * taken once, at the last diagonal (:1556) */
            if (P.unbanded ? calcs++ == 0 : calcs++ % 10 == 0) {
                /* diagonalCalculationTotalProbability :736-754, by wave 0 */
                if (wave == 0) {
                    double acc = CP_NEG_INF;
                    for (int base = 0; base < w2; base += 64) {
                        int cc = base + lane;
                        bool valid = cc < w2;
                        double v = CP_NEG_INF;
                        if (valid) { /* cell_dotProduct :391-397 */
                            v = fdd[cc * 3] + bdd[cc * 3];
                            v = cp_logAdd(v, fdd[cc * 3 + 1] + bdd[cc * 3 + 1]);
                            v = cp_logAdd(v, fdd[cc * 3 + 2] + bdd[cc * 3 + 2]);
                        }
                        acc = cp_wave_seq_fold(acc, v, valid);
                    }
                    if (d2 + 1 <= dTop) {
                        /* matches that step over d2: forward[d2-1] --match--> cells of d2+1 */
                        const int l3 = c.L[d2 + 1], w3 = ((c.R[d2 + 1] - l3) >> 1) + 1;
                        const double *b3 = bslot(c, d2 + 1);
                        double acc2 = CP_NEG_INF;
                        for (int base = 0; base < w3; base += 64) {
                            int cc = base + lane;
                            bool valid = cc < w3;
                            double v = CP_NEG_INF;
                            if (valid) {
                                int xmy = l3 + 2 * cc;
                                const double *mid = fcell(c, d2 - 1, xmy);
                                double mm = CP_NEG_INF;
                                if (mid) {
                                    long long x = (d2 + 1 + xmy) / 2, y = (d2 + 1 - xmy) / 2;
                                    const double *row = row_of(c, x - 1);
                                    double mean, noise;
                                    event_of(c, y - 1, mean, noise);
                                    double eP = emit_match(row, mean, noise);
                                    mm = cp_logAdd(mm, mid[0] + (eP + t[T_MATCH_CONTINUE]));
                                    mm = cp_logAdd(mm, mid[1] + (eP + t[T_MATCH_FROM_GAP_X]));
                                    mm = cp_logAdd(mm, mid[2] + (eP + t[T_MATCH_FROM_GAP_Y]));
                                }
                                v = mm + b3[cc * 3];
                                v = cp_logAdd(v, CP_NEG_INF + b3[cc * 3 + 1]);
                                v = cp_logAdd(v, CP_NEG_INF + b3[cc * 3 + 2]);
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
                double *o = dbgB + (it.cellBase + c.pre[d2]) * 3;
                for (int cc = tid; cc < w2 * 3; cc += 256) o[cc] = bdd[cc];
            }

            if (P.mode == 0) {
                /* diagonalCalculationPosteriorMatchProbs :756-795, ordered emission by wave 0 */
                if (wave == 0) {
                    for (int base = 0; base < w2; base += 64) {
                        int cc = base + lane;
                        bool hit = false;
                        double e = 0.0, p = 0.0;
                        long long x = 0, y = 0;
                        if (cc < w2) {
                            int xmy = l2 + 2 * cc;
                            x = (d2 + xmy) / 2; y = (d2 - xmy) / 2;
                            if (x > 0 && y > 0) {
                                e = (fdd[cc * 3] + bdd[cc * 3]) - total;
                                p = exp(e);
                                hit = p >= P.threshold;
                            }
                        }
                        unsigned long long m = __ballot(hit);
                        if (hit) {
                            long long idx = myPairs + __popcll(m & ((1ull << lane) - 1ull));
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
            } else {
                /* diagonalCalculation_Expectations :841-863 with
                 * cell_signal_updateTransAndKmerSkipExpectations :426-443 */
                if (tid == 0) expAcc[9] += total;
                const bool haveMiddle = d2 - 2 >= tracedBackTo; /* forward[d2-2] already freed otherwise */
                for (int cc = tid; cc < w2; cc += 256) {
                    int xmy = l2 + 2 * cc;
                    long long x = (d2 + xmy) / 2, y = (d2 - xmy) / 2;
                    const double *cur = bdd + cc * 3;
                    const double *row = row_of(c, x - 1);
                    double mean, noise;
                    event_of(c, y - 1, mean, noise);
                    const double *lower = fcell(c, d2 - 1, xmy - 1);
                    const double *middle = haveMiddle ? fcell(c, d2 - 2, xmy) : nullptr;
                    const double *upper = fcell(c, d2 - 1, xmy + 1);
                    if (lower) {
                        double eP = row[CP_GAPX];
                        double p0 = exp(lower[0] + cur[1] + (eP + t[T_GAP_OPEN_X]) - total);
                        double p1 = exp(lower[1] + cur[1] + (eP + t[T_GAP_EXTEND_X]) - total);
                        double p2 = exp(lower[2] + cur[1] + (eP + t[T_GAP_SWITCH_TO_X]) - total);
                        expAcc[0 * 3 + 1] += p0;
                        expAcc[1 * 3 + 1] += p1;
                        expAcc[2 * 3 + 1] += p2;
                        int k = x - 1 >= 0 ? (int) c.kidx[x - 1] : 4096;
                        if (k < 4096 && gapAcc) {
                            atomicAdd(gapAcc + k, p0);
                            atomicAdd(gapAcc + k, p1);
                            atomicAdd(gapAcc + k, p2);
                        }
                    }
                    if (middle) {
                        double eP = emit_match(row, mean, noise);
                        expAcc[0 * 3 + 0] += exp(middle[0] + cur[0] + (eP + t[T_MATCH_CONTINUE]) - total);
                        expAcc[1 * 3 + 0] += exp(middle[1] + cur[0] + (eP + t[T_MATCH_FROM_GAP_X]) - total);
                        expAcc[2 * 3 + 0] += exp(middle[2] + cur[0] + (eP + t[T_MATCH_FROM_GAP_Y]) - total);
                    }
                    if (upper) {
                        double eP = emit_gapy(row, mean, noise);
                        expAcc[0 * 3 + 2] += exp(upper[0] + cur[2] + (eP + t[T_GAP_OPEN_Y]) - total);
                        expAcc[2 * 3 + 2] += exp(upper[2] + cur[2] + (eP + t[T_GAP_EXTEND_Y]) - total);
                    }
                }
            }
            __syncthreads();
        }
        tracedBackTo = tracedBackFrom;
    }

    if (P.mode == 1 && expect) {
        /* block reduction of the per-thread partial sums, then one atomic per value */
        if (tid < 16) sExp[tid] = 0.0;
        __syncthreads();
        for (int i = 0; i < 10; i++) {
            double v = expAcc[i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if (lane == 0) atomicAdd(&sExp[i], v);
        }
        __syncthreads();
        double *dst = expect + (long long) it.model * (9 + 4096 + 1);
        if (tid < 9) atomicAdd(dst + tid, sExp[tid]);
        if (tid == 9) atomicAdd(dst + 9 + 4096, sExp[9]);
    }
    if (tid == 0) {
        nPairs[blockIdx.x] = myPairs;
        nTot[blockIdx.x] = myTot;
    }
}

/* k-mer index of every position of the concatenated nucleotide buffer
 * (emissions_discrete_getKmerIndex impl/stateMachine.c:104-139): A,C,G,T = 0..3, most significant
 * first; any other character makes the 6-mer "not a k-mer" (index 4096 here, > 4096 there). */
extern "C" __global__ void cpecan_k_kmer_index(const char *chars, long long n, unsigned short *kidx) {
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int v = 0;
    bool ok = i + 5 < n;
    if (ok) {
        for (int j = 0; j < 6; j++) {
            char ch = chars[i + j];
            int b = ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1;
            if (b < 0) ok = false;
            v = v * 4 + (b & 3);
        }
    }
    kidx[i] = ok ? (unsigned short) v : (unsigned short) 4096;
}

/*
 * cpecan_kernel_generalh.hip -- banded forward / backward / posterior DP for the reference's 3-state
 * HDP signal machine (stateMachine3HDP_cellCalculate, impl/stateMachine.c:1336-1366; SURVEY R13,
 * BASELINE configs[4]): the 3-state transitions of the strawMan machine, X-gap emission log(0.1),
 * match and Y-gap emission = the posterior-predictive density of the k-mer's Dirichlet process at the
 * event mean (get_nanopore_kmer_density impl/nanopore_hdp.c:390 -> dir_proc_density impl/hdp.c:2577
 * -> grid_spline_interp impl/hdp_math_utils.c:471) -- a linear density used where a log-probability
 * is expected, exactly as the reference does (quirk Q6).  X elements are read as sequence_getKmer3
 * does (impl/pairwiseAligner.c:327-331).
 *
 * Structure of cpecan_kernel_general.hip: one 256-thread workgroup per work item, any band width,
 * forward diagonals in HBM, three rotating backward diagonals, posterior decode only.  The spline
 * tables (values and slopes on the sampling grid, one row per OBSERVED Dirichlet process) stay in
 * HBM; a cell gathers four doubles from the row of its k-mer's nearest observed ancestor, which the
 * host resolved per k-mer id when the model was uploaded.
 */
#include "cpecan_device.h"


namespace {

struct CtxH {
    const int *L, *R;
    const long long *pre;
    const int *kid;     /* k-mer id (over the model's alphabet) per X character position; -1: bad character */
    const double *ev;   /* events, 3 doubles each */
    DevHdpModel m;
    double *F, *Bws;
    int maxWidth;
};

/* grid_spline_interp (evenly spaced grid), then the clamp of dir_proc_density */
__device__ __forceinline__ double density(const CtxH &c, long long ix, long long iy) {
    const int id = c.kid[ix >= 0 ? ix : 0]; /* sequence_getKmer3: index < 0 reads the first k-mer */
    const double q = iy >= 0 ? c.ev[3 * iy] : CP_NEG_INF; /* NULLEVENT mean */
    if (id < 0) return q - q; /* NaN: the reference exits on a character outside the alphabet */
    const long long row = c.m.kmerRow[id];
    const double *x = c.m.grid, *y = c.m.y + row * c.m.gridLength, *s = c.m.slope + row * c.m.gridLength;
    const int n = c.m.gridLength - 1;
    double r;
    if (q <= x[0]) r = y[0] - s[0] * (x[0] - q);
    else if (q >= x[n]) r = y[n] + s[n] * (q - x[n]);
    else {
        const double dx = x[1] - x[0];
        const long long il = (long long) ((q - x[0]) / dx), ir = il + 1;
        const double dy = y[ir] - y[il];
        const double a = s[il] * dx - dy;
        const double b = dy - s[ir] * dx;
        const double tl = (q - x[il]) / dx;
        const double tr = 1.0 - tl;
        r = tr * y[il] + tl * y[ir] + tl * tr * (a * tr + b * tl);
    }
    return r > 0.0 ? r : 0.0;
}

__device__ __forceinline__ const double *fcellh(const CtxH &c, long long d, int xmy) {
    if (d < 0) return nullptr;
    const int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    return c.F + (c.pre[d] + ((xmy - l) >> 1)) * 3;
}
__device__ __forceinline__ double *bsloth(const CtxH &c, long long d) {
    return c.Bws + (d % 3) * (long long) c.maxWidth * 3;
}
__device__ __forceinline__ const double *bcellh(const CtxH &c, long long d, long long dTop, int xmy) {
    if (d > dTop) return nullptr;
    const int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    return bsloth(c, d) + ((xmy - l) >> 1) * 3;
}

#define GAPX_EP (-2.3025850929940455) /* log(0.1), stateMachine.c:1347 */

__device__ __forceinline__ double match_fromh(const double *middle, double eP, const double *t) {
    double m = CP_NEG_INF;
    m = cp_logAdd(m, middle[0] + (eP + t[T_MATCH_CONTINUE]));
    m = cp_logAdd(m, middle[1] + (eP + t[T_MATCH_FROM_GAP_X]));
    m = cp_logAdd(m, middle[2] + (eP + t[T_MATCH_FROM_GAP_Y]));
    return m;
}

__device__ __forceinline__ void forward_cellh(const CtxH &c, long long d, int xmy, double o[3]) {
    const long long x = (d + xmy) / 2, y = (d - xmy) / 2;
    const double *t = c.m.t;
    o[0] = o[1] = o[2] = CP_NEG_INF;
    const double *lower = fcellh(c, d - 1, xmy - 1);
    const double *middle = fcellh(c, d - 2, xmy);
    const double *upper = fcellh(c, d - 1, xmy + 1);
    if (lower) {
        o[1] = cp_logAdd(o[1], lower[0] + (GAPX_EP + t[T_GAP_OPEN_X]));
        o[1] = cp_logAdd(o[1], lower[1] + (GAPX_EP + t[T_GAP_EXTEND_X]));
        o[1] = cp_logAdd(o[1], lower[2] + (GAPX_EP + t[T_GAP_SWITCH_TO_X]));
    }
    if (middle) o[0] = match_fromh(middle, density(c, x - 1, y - 1), t);
    if (upper) {
        const double eP = density(c, x - 1, y - 1);
        o[2] = cp_logAdd(o[2], upper[0] + (eP + t[T_GAP_OPEN_Y]));
        o[2] = cp_logAdd(o[2], upper[2] + (eP + t[T_GAP_EXTEND_Y]));
    }
}

/* gather form of cell_calculateBackward, the reference's scatter order kept per target state */
__device__ __forceinline__ void backward_cellh(const CtxH &c, long long d, long long dTop, int xmy, double o[3]) {
    const long long x = (d + xmy) / 2, y = (d - xmy) / 2;
    const double *t = c.m.t;
    o[0] = o[1] = o[2] = CP_NEG_INF;
    const double *s2 = bcellh(c, d + 2, dTop, xmy);
    if (s2) {
        const double eP = density(c, x, y);
        o[0] = cp_logAdd(o[0], s2[0] + (eP + t[T_MATCH_CONTINUE]));
        o[1] = cp_logAdd(o[1], s2[0] + (eP + t[T_MATCH_FROM_GAP_X]));
        o[2] = cp_logAdd(o[2], s2[0] + (eP + t[T_MATCH_FROM_GAP_Y]));
    }
    const double *su = bcellh(c, d + 1, dTop, xmy - 1);
    if (su) {
        const double eP = density(c, x - 1, y);
        o[0] = cp_logAdd(o[0], su[2] + (eP + t[T_GAP_OPEN_Y]));
        o[2] = cp_logAdd(o[2], su[2] + (eP + t[T_GAP_EXTEND_Y]));
    }
    const double *sl = bcellh(c, d + 1, dTop, xmy + 1);
    if (sl) {
        o[0] = cp_logAdd(o[0], sl[1] + (GAPX_EP + t[T_GAP_OPEN_X]));
        o[1] = cp_logAdd(o[1], sl[1] + (GAPX_EP + t[T_GAP_EXTEND_X]));
        o[2] = cp_logAdd(o[2], sl[1] + (GAPX_EP + t[T_GAP_SWITCH_TO_X]));
    }
}

} // namespace

extern "C" __global__ __launch_bounds__(256) void cpecan_k_generalh(
    const DevItem *items, DevParams P, const int *bandL, const int *bandR,
    const long long *cellPrefix, const int *kid, const double *events,
    const DevHdpModel *models, double *Fstore, double *Bstore, long long *pairs,
    double *pairLogp, long long *nPairs, long long *totXay, double *totVal, long long *nTot, double *expect) {
    const DevItem it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    CtxH c;
    c.L = bandL + it.diagBase;
    c.R = bandR + it.diagBase;
    c.pre = cellPrefix + it.diagBase;
    c.kid = kid + it.xOff;
    c.ev = events + 3 * it.yOff;
    c.m = models[it.model];
    const double *t = c.m.t;
    c.F = Fstore + it.cellBase * 3;
    c.Bws = Bstore + it.bwsBase;
    c.maxWidth = it.maxWidth;

    __shared__ double sTotal;
    __shared__ double sExp[16];
    double expAcc[10]; /* per-thread partial expectations: 9 transitions + likelihood */
    for (int i = 0; i < 10; i++) expAcc[i] = 0.0;
    const long long D = it.lX + it.lY;
    long long myPairs = 0, myTot = 0;
    if (D == 0) {
        if (tid == 0) { nPairs[blockIdx.x] = 0; nTot[blockIdx.x] = 0; }
        return;
    }
    /* diagonal 0: stateMachine3_startStateProb / raggedStartStateProb (:1168-1177), shared with sm3 */
    if (tid == 0) {
        c.F[0] = it.raggedL ? CP_NEG_INF : 0.0;
        c.F[1] = it.raggedL ? 0.0 : CP_NEG_INF;
        c.F[2] = it.raggedL ? 0.0 : CP_NEG_INF;
    }
    __threadfence_block();
    __syncthreads();

    long long tracedBackTo = 0;
    for (long long d = 1; d <= D; d++) {
        const int l = c.L[d], width = ((c.R[d] - l) >> 1) + 1;
        double *fd = c.F + c.pre[d] * 3;
        for (int cc = tid; cc < width; cc += 256) {
            double o[3];
            forward_cellh(c, d, l + 2 * cc, o);
            fd[cc * 3] = o[0]; fd[cc * 3 + 1] = o[1]; fd[cc * 3 + 2] = o[2];
        }
        __threadfence_block();
        __syncthreads();

        const bool atEnd = d == D;
        const bool tb = !P.unbanded && d >= tracedBackTo + P.minDiags && width <= P.expansion * 2 + 1;
        if (!(atEnd || tb)) continue;

        const long long dTop = d;
        const long long tracedBackFrom = dTop - (atEnd ? 0 : P.tbDiags + 1);
        {
            double e0, e1, e2; /* stateMachine3_endStateProb / raggedEndStateProb (:1179-1207) */
            if (atEnd && it.raggedR) {
                e0 = (t[T_GAP_OPEN_X] + t[T_GAP_OPEN_Y]) / 2.0;
                e1 = t[T_GAP_EXTEND_X];
                e2 = t[T_GAP_EXTEND_Y];
            } else {
                e0 = t[T_MATCH_CONTINUE];
                e1 = t[T_MATCH_FROM_GAP_X];
                e2 = t[T_MATCH_FROM_GAP_Y];
            }
            double *b = bsloth(c, dTop);
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
                double *b = bsloth(c, d2);
                for (int cc = tid; cc < w2; cc += 256) {
                    double o[3];
                    backward_cellh(c, d2, dTop, l2 + 2 * cc, o);
                    b[cc * 3] = o[0]; b[cc * 3 + 1] = o[1]; b[cc * 3 + 2] = o[2];
                }
                __threadfence_block();
                __syncthreads();
            }
            if (d2 > tracedBackFrom) continue;

            const double *fdd = c.F + c.pre[d2] * 3;
            const double *bdd = bsloth(c, d2);
            if (P.unbanded ? calcs++ == 0 : calcs++ % 10 == 0) {
                /* diagonalCalculationTotalProbability :736-754, by wave 0 */
                if (wave == 0) {
                    double acc = CP_NEG_INF;
                    for (int base = 0; base < w2; base += 64) {
                        const int cc = base + lane;
                        const bool valid = cc < w2;
                        double v = CP_NEG_INF;
                        if (valid) {
                            v = fdd[cc * 3] + bdd[cc * 3];
                            v = cp_logAdd(v, fdd[cc * 3 + 1] + bdd[cc * 3 + 1]);
                            v = cp_logAdd(v, fdd[cc * 3 + 2] + bdd[cc * 3 + 2]);
                        }
                        acc = cp_wave_seq_fold(acc, v, valid);
                    }
                    if (d2 + 1 <= dTop) {
                        const int l3 = c.L[d2 + 1], w3 = ((c.R[d2 + 1] - l3) >> 1) + 1;
                        const double *b3 = bsloth(c, d2 + 1);
                        double acc2 = CP_NEG_INF;
                        for (int base = 0; base < w3; base += 64) {
                            const int cc = base + lane;
                            const bool valid = cc < w3;
                            double v = CP_NEG_INF;
                            if (valid) {
                                const int xmy = l3 + 2 * cc;
                                const double *mid = fcellh(c, d2 - 1, xmy);
                                double mm = CP_NEG_INF;
                                if (mid) {
                                    const long long x = (d2 + 1 + xmy) / 2, y = (d2 + 1 - xmy) / 2;
                                    mm = match_fromh(mid, density(c, x - 1, y - 1), t);
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

            if (P.mode == 1) {
                /* diagonalCalculation_Expectations :841-863 with
                 * cell_signal_updateTransAndKmerSkipExpectations2 :445-476: every transition adds its posterior
                 * to the transition counts; one INTO match with posterior >= the HdpHmm's threshold (carried in
                 * P.threshold) also assigns the cell's event to its k-mer.  Assignments come out in the host
                 * loop's order (cells by x-y, per cell from match, gapX, gapY) as (from, x-1, y-1) triples. */
                if (tid == 0) expAcc[9] += total;
                const bool haveMiddle = d2 - 2 >= tracedBackTo; /* forward[d2-2] is freed otherwise (:982) */
                for (int cc = tid; cc < w2; cc += 256) {
                    const int xmy = l2 + 2 * cc;
                    const long long x = (d2 + xmy) / 2, y = (d2 - xmy) / 2;
                    const double *cur = bdd + cc * 3;
                    const double *lower = fcellh(c, d2 - 1, xmy - 1);
                    const double *middle = haveMiddle ? fcellh(c, d2 - 2, xmy) : nullptr;
                    const double *upper = fcellh(c, d2 - 1, xmy + 1);
                    if (lower) {
                        expAcc[0 * 3 + 1] += exp(lower[0] + cur[1] + (GAPX_EP + t[T_GAP_OPEN_X]) - total);
                        expAcc[1 * 3 + 1] += exp(lower[1] + cur[1] + (GAPX_EP + t[T_GAP_EXTEND_X]) - total);
                        expAcc[2 * 3 + 1] += exp(lower[2] + cur[1] + (GAPX_EP + t[T_GAP_SWITCH_TO_X]) - total);
                    }
                    if (middle) {
                        const double eP = density(c, x - 1, y - 1);
                        expAcc[0 * 3 + 0] += exp(middle[0] + cur[0] + (eP + t[T_MATCH_CONTINUE]) - total);
                        expAcc[1 * 3 + 0] += exp(middle[1] + cur[0] + (eP + t[T_MATCH_FROM_GAP_X]) - total);
                        expAcc[2 * 3 + 0] += exp(middle[2] + cur[0] + (eP + t[T_MATCH_FROM_GAP_Y]) - total);
                    }
                    if (upper) {
                        const double eP = density(c, x - 1, y - 1);
                        expAcc[0 * 3 + 2] += exp(upper[0] + cur[2] + (eP + t[T_GAP_OPEN_Y]) - total);
                        expAcc[2 * 3 + 2] += exp(upper[2] + cur[2] + (eP + t[T_GAP_EXTEND_Y]) - total);
                    }
                }
                if (wave == 0 && haveMiddle) {
                    for (int base = 0; base < w2; base += 64) {
                        const int cc = base + lane;
                        double e[3] = { 0.0, 0.0, 0.0 };
                        bool hit[3] = { false, false, false };
                        long long x = 0, y = 0;
                        if (cc < w2) {
                            const int xmy = l2 + 2 * cc;
                            x = (d2 + xmy) / 2;
                            y = (d2 - xmy) / 2;
                            const double *middle = fcellh(c, d2 - 2, xmy);
                            if (middle) {
                                const double eP = density(c, x - 1, y - 1), cm = bdd[cc * 3];
                                e[0] = middle[0] + cm + (eP + t[T_MATCH_CONTINUE]) - total;
                                e[1] = middle[1] + cm + (eP + t[T_MATCH_FROM_GAP_X]) - total;
                                e[2] = middle[2] + cm + (eP + t[T_MATCH_FROM_GAP_Y]) - total;
#pragma unroll
                                for (int f = 0; f < 3; f++) hit[f] = exp(e[f]) >= P.threshold;
                            }
                        }
                        const unsigned long long below = (1ull << lane) - 1ull;
                        const unsigned long long m0 = __ballot(hit[0]), m1 = __ballot(hit[1]), m2 = __ballot(hit[2]);
                        long long idx = myPairs + __popcll(m0 & below) + __popcll(m1 & below) + __popcll(m2 & below);
#pragma unroll
                        for (int f = 0; f < 3; f++) {
                            if (!hit[f]) continue;
                            if (idx < it.pairCap) {
                                long long *o = pairs + (it.pairBase + idx) * 3;
                                o[0] = f;
                                o[1] = x - 1;
                                o[2] = y - 1;
                                pairLogp[it.pairBase + idx] = e[f];
                            }
                            idx++;
                        }
                        myPairs += __popcll(m0) + __popcll(m1) + __popcll(m2);
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
                            e = (fdd[cc * 3] + bdd[cc * 3]) - total;
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
        /* block reduction of the per-thread partial sums, then one atomic per value */
        if (tid < 16) sExp[tid] = 0.0;
        __syncthreads();
        for (int i = 0; i < 10; i++) {
            double v = expAcc[i];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if (lane == 0) atomicAdd(&sExp[i], v);
        }
        __syncthreads();
        if (tid < 10) atomicAdd(expect + (long long) it.model * 10 + tid, sExp[tid]);
    }
    if (tid == 0) {
        nPairs[blockIdx.x] = myPairs;
        nTot[blockIdx.x] = myTot;
    }
}

/* k-mer id over the model's alphabet for every position of the concatenated nucleotide buffer
 * (kmer_id impl/nanopore_hdp.c:348-380: most significant character first); -1 where one of the six
 * characters is outside the alphabet (the reference exits there) or the buffer ends */
extern "C" __global__ void cpecan_k_hdp_kmer_id(const char *chars, long long n, unsigned long long alphabet,
                                                unsigned long long alphabetHi, int alphabetSize, int *kid) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int v = 0;
    bool ok = i + 5 < n;
    if (ok) {
        for (int j = 0; j < 6; j++) {
            const char ch = chars[i + j];
            int d = -1;
            for (int a = 0; a < alphabetSize; a++) {
                const char ac = (char) ((a < 8 ? alphabet >> (8 * a) : alphabetHi >> (8 * (a - 8))) & 0xff);
                if (ac == ch) d = a;
            }
            if (d < 0) ok = false;
            v = v * alphabetSize + (d < 0 ? 0 : d);
        }
    }
    kid[i] = ok ? v : -1;
}

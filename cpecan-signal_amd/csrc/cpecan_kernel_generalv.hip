/*
 * cpecan_kernel_generalv.hip -- banded forward / backward / posterior DP for the reference's
 * 3-state "vanilla" signal machine (stateMachine3Vanilla_cellCalculate, impl/stateMachine.c:1368-1409;
 * SURVEY R12): transition probabilities that depend on the reference position through 30 skip bins
 * (emissions_signal_getBetaOrAlphaSkipProb :421, getKmerSkipBin :388), Gaussian level + inverse-
 * Gaussian noise emissions (emissions_signal_getEventMatchProbWithTwoDists :499-528,
 * logInvGaussPdf :322-331), X elements read as sequence_getKmer2 does (impl/pairwiseAligner.c:320-325).
 *
 * Structure of cpecan_kernel_general.hip: one 256-thread workgroup per work item, any band width,
 * forward diagonals in HBM, three rotating backward diagonals, posterior decode only.
 *
 * Every log() the reference takes per cell is a function of the skip bin, the k-mer or the event
 * alone, so the host takes them once with its libm (cpecan_hip.hip: derive_vanilla); the device
 * adds them in the reference's order.
 */
#include "cpecan_device.h"

namespace {

struct CtxV {
    const int *L, *R;
    const long long *pre;
    const unsigned short *kidx; /* k-mer index per X character position (4096 = not a k-mer) */
    const double *ev;           /* events, 3 doubles each */
    const double *lnoise;       /* log(event noise), host libm */
    const double *hdr;          /* model header: scalars and per-bin log transition probabilities */
    const double *rows;         /* CP_VROW doubles per k-mer */
    double *F, *Bws;
    int maxWidth;
};

/* the two k-mers sequence_getKmer2 exposes for sequence index ix: a pointer to character
 * max(ix-1, 0); the skip bin looks at the k-mers at +0 and +1, the emission at the one at +1
 * (so sequence index 0 is scored with k-mer 1, as in the reference) */
__device__ __forceinline__ void kmers_of(const CtxV &c, long long ix, int &kPrev, int &kCur) {
    const long long p = ix > 0 ? ix - 1 : 0;
    kPrev = c.kidx[p];
    kCur = c.kidx[p + 1];
}
/* per-bin log transition probabilities: [bin][log a_mx, log a_xx, log a_mm, log a_xm, log a_my] */
__device__ __forceinline__ const double *bin_logs(const CtxV &c, int kPrev, int kCur) {
    const double d = fabs(c.rows[(long long) kCur * CP_VROW + CP_V_MU] - c.rows[(long long) kPrev * CP_VROW + CP_V_MU]);
    long long bin = (long long) (d / 0.5);
    if (bin >= 30) bin = 29;
    return c.hdr + CP_VHDR_BINS + bin * 5;
}
/* emissions_signal_getEventMatchProbWithTwoDists on table `o` (0: match table, 6: extra-event table) */
__device__ __forceinline__ double emit2(const CtxV &c, int k, long long iy, int o) {
    const double *r = c.rows + (long long) k * CP_VROW + o;
    double mean, noise, lnoise;
    if (iy >= 0) {
        mean = c.ev[3 * iy];
        noise = c.ev[3 * iy + 1];
        lnoise = c.lnoise[iy];
    } else { /* NULLEVENT {-inf, 0} (:261): log(0) = -inf */
        mean = CP_NEG_INF;
        noise = 0.0;
        lnoise = CP_NEG_INF;
    }
    const double level = cp_logGauss(mean, r[CP_V_MU], r[CP_V_SD], r[CP_V_K]);
    const double a = (noise - r[CP_V_NMU]) / r[CP_V_NMU];
    const double l_twoPi = 1.8378770664093453;
    const double nz = (r[CP_V_LLAMBDA] - l_twoPi - 3 * lnoise - r[CP_V_LAMBDA] * a * a / noise) / 2;
    return level + nz;
}

__device__ __forceinline__ const double *fcellv(const CtxV &c, long long d, int xmy) {
    if (d < 0) return nullptr;
    const int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    return c.F + (c.pre[d] + ((xmy - l) >> 1)) * 3;
}
__device__ __forceinline__ double *bslotv(const CtxV &c, long long d) {
    return c.Bws + (d % 3) * (long long) c.maxWidth * 3;
}
__device__ __forceinline__ const double *bcellv(const CtxV &c, long long d, long long dTop, int xmy) {
    if (d > dTop) return nullptr;
    const int l = c.L[d], r = c.R[d];
    if (xmy < l || xmy > r) return nullptr;
    return bslotv(c, d) + ((xmy - l) >> 1) * 3;
}

__device__ __forceinline__ double match_fromv(const double *middle, double eP, const double *bl, const double *hdr) {
    double m = CP_NEG_INF;
    m = cp_logAdd(m, middle[0] + (eP + bl[2]));               /* log a_mm */
    m = cp_logAdd(m, middle[1] + (eP + bl[3]));               /* log a_xm */
    m = cp_logAdd(m, middle[2] + (eP + hdr[CP_VHDR_LOG_YM])); /* log a_ym */
    return m;
}

/* cell_calculateForward over stateMachine3Vanilla_cellCalculate */
__device__ __forceinline__ void forward_cellv(const CtxV &c, long long d, int xmy, double o[3]) {
    const long long x = (d + xmy) / 2, y = (d - xmy) / 2;
    int kPrev, kCur;
    kmers_of(c, x - 1, kPrev, kCur);
    const double *bl = bin_logs(c, kPrev, kCur);
    o[0] = o[1] = o[2] = CP_NEG_INF;
    const double *lower = fcellv(c, d - 1, xmy - 1);
    const double *middle = fcellv(c, d - 2, xmy);
    const double *upper = fcellv(c, d - 1, xmy + 1);
    if (lower) {
        o[1] = cp_logAdd(o[1], lower[0] + (0 + bl[0])); /* log a_mx */
        o[1] = cp_logAdd(o[1], lower[1] + (0 + bl[1])); /* log a_xx */
    }
    if (middle) o[0] = match_fromv(middle, emit2(c, kCur, y - 1, 0), bl, c.hdr);
    if (upper) {
        const double eP = emit2(c, kCur, y - 1, 6);
        o[2] = cp_logAdd(o[2], upper[0] + (eP + bl[4]));                   /* log a_my */
        o[2] = cp_logAdd(o[2], upper[2] + (eP + c.hdr[CP_VHDR_LOG_YY])); /* log a_yy */
    }
}

/* gather form of cell_calculateBackward, the reference's scatter order kept per target state */
__device__ __forceinline__ void backward_cellv(const CtxV &c, long long d, long long dTop, int xmy, double o[3]) {
    const long long x = (d + xmy) / 2, y = (d - xmy) / 2;
    o[0] = o[1] = o[2] = CP_NEG_INF;
    /* (ii) cell (x+1, y+1) on d+2, its middle block: its X element is index x, its event index y */
    const double *s2 = bcellv(c, d + 2, dTop, xmy);
    if (s2) {
        int kPrev, kCur;
        kmers_of(c, x, kPrev, kCur);
        const double *bl = bin_logs(c, kPrev, kCur);
        const double eP = emit2(c, kCur, y, 0);
        o[0] = cp_logAdd(o[0], s2[0] + (eP + bl[2]));
        o[1] = cp_logAdd(o[1], s2[0] + (eP + bl[3]));
        o[2] = cp_logAdd(o[2], s2[0] + (eP + c.hdr[CP_VHDR_LOG_YM]));
    }
    /* (iii) cell (x, y+1) on d+1, its upper block: X element x-1, event y */
    const double *su = bcellv(c, d + 1, dTop, xmy - 1);
    if (su) {
        int kPrev, kCur;
        kmers_of(c, x - 1, kPrev, kCur);
        const double *bl = bin_logs(c, kPrev, kCur);
        const double eP = emit2(c, kCur, y, 6);
        o[0] = cp_logAdd(o[0], su[2] + (eP + bl[4]));
        o[2] = cp_logAdd(o[2], su[2] + (eP + c.hdr[CP_VHDR_LOG_YY]));
    }
    /* (iv) cell (x+1, y) on d+1, its lower block: X element x */
    const double *sl = bcellv(c, d + 1, dTop, xmy + 1);
    if (sl) {
        int kPrev, kCur;
        kmers_of(c, x, kPrev, kCur);
        const double *bl = bin_logs(c, kPrev, kCur);
        o[0] = cp_logAdd(o[0], sl[1] + (0 + bl[0]));
        o[1] = cp_logAdd(o[1], sl[1] + (0 + bl[1]));
    }
}

} // namespace

extern "C" __global__ __launch_bounds__(256) void cpecan_k_generalv(
    const DevItem *items, DevParams P, const int *bandL, const int *bandR,
    const long long *cellPrefix, const unsigned short *kidx, const double *events,
    const double *logNoise, const double *models, double *Fstore, double *Bstore, long long *pairs,
    double *pairLogp, long long *nPairs, long long *totXay, double *totVal, long long *nTot, double *expect) {
    const DevItem it = items[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    CtxV c;
    c.L = bandL + it.diagBase;
    c.R = bandR + it.diagBase;
    c.pre = cellPrefix + it.diagBase;
    c.kidx = kidx + it.xOff;
    c.ev = events + 3 * it.yOff;
    c.lnoise = logNoise + it.yOff;
    c.hdr = models + (long long) it.model * CP_VMODEL_STRIDE;
    c.rows = c.hdr + CP_VHDR;
    c.F = Fstore + it.cellBase * 3;
    c.Bws = Bstore + it.bwsBase;
    c.maxWidth = it.maxWidth;

    __shared__ double sTotal;
    /* Baum-Welch sums of this alignment (VanillaHmm): 30 beta + 30 alpha skip bins and the likelihood,
     * one copy per wave, folded into the model's block of `expect` at the end */
    __shared__ double sExp[4][CP_EXPECTV_LEN + 1];
    for (int i = tid; i < 4 * (CP_EXPECTV_LEN + 1); i += 256) (&sExp[0][0])[i] = 0.0;
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
            forward_cellv(c, d, l + 2 * cc, o);
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
            /* stateMachine3Vanilla_endStateProb / raggedEndStateProb (:1209-1235) */
            const double eM = c.hdr[CP_VHDR_END_M], eX = c.hdr[CP_VHDR_END_X], eY = c.hdr[CP_VHDR_END_Y];
            const double e0 = atEnd && it.raggedR ? (eX + eY) / 2.0 : eM;
            double *b = bslotv(c, dTop);
            for (int cc = tid; cc < width; cc += 256) {
                b[cc * 3] = e0; b[cc * 3 + 1] = eX; b[cc * 3 + 2] = eY;
            }
        }
        __threadfence_block();
        __syncthreads();

        double total = CP_NEG_INF;
        long long calcs = 0;
        for (long long d2 = dTop; d2 > tracedBackTo; d2--) {
            const int l2 = c.L[d2], w2 = ((c.R[d2] - l2) >> 1) + 1;
            if (d2 < dTop) {
                double *b = bslotv(c, d2);
                for (int cc = tid; cc < w2; cc += 256) {
                    double o[3];
                    backward_cellv(c, d2, dTop, l2 + 2 * cc, o);
                    b[cc * 3] = o[0]; b[cc * 3 + 1] = o[1]; b[cc * 3 + 2] = o[2];
                }
                __threadfence_block();
                __syncthreads();
            }
            if (d2 > tracedBackFrom) continue;

            const double *fdd = c.F + c.pre[d2] * 3;
            const double *bdd = bslotv(c, d2);
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
                        const double *b3 = bslotv(c, d2 + 1);
                        double acc2 = CP_NEG_INF;
                        for (int base = 0; base < w3; base += 64) {
                            const int cc = base + lane;
                            const bool valid = cc < w3;
                            double v = CP_NEG_INF;
                            if (valid) {
                                const int xmy = l3 + 2 * cc;
                                const double *mid = fcellv(c, d2 - 1, xmy);
                                double mm = CP_NEG_INF;
                                if (mid) {
                                    const long long x = (d2 + 1 + xmy) / 2, y = (d2 + 1 - xmy) / 2;
                                    int kPrev, kCur;
                                    kmers_of(c, x - 1, kPrev, kCur);
                                    mm = match_fromv(mid, emit2(c, kCur, y - 1, 0), bin_logs(c, kPrev, kCur), c.hdr);
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
                /* diagonalCalculation_Expectations :841-863 with cell_signal_updateBetaAndAlphaProb :478-498:
                 * of all transitions only match->gapX (into the cell's skip bin) and gapX->gapX (bin + 30)
                 * are collected; both live in the lower block */
                double *acc = sExp[wave];
                if (tid == 0) acc[CP_EXPECTV_LEN - 1] += total;
                for (int cc = tid; cc < w2; cc += 256) {
                    const int xmy = l2 + 2 * cc;
                    const long long x = (d2 + xmy) / 2;
                    const double *lower = fcellv(c, d2 - 1, xmy - 1);
                    if (!lower) continue;
                    int kPrev, kCur;
                    kmers_of(c, x - 1, kPrev, kCur);
                    const double *bl = bin_logs(c, kPrev, kCur);
                    const int bin = (int) ((bl - (c.hdr + CP_VHDR_BINS)) / 5);
                    const double *cur = bdd + cc * 3;
                    atomicAdd(&acc[bin], exp(lower[0] + cur[1] + (0 + bl[0]) - total));
                    atomicAdd(&acc[bin + 30], exp(lower[1] + cur[1] + (0 + bl[1]) - total));
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
        __syncthreads();
        double *dst = expect + (long long) it.model * CP_EXPECTV_LEN;
        for (int i = tid; i < CP_EXPECTV_LEN; i += 256) {
            const double v = ((sExp[0][i] + sExp[1][i]) + sExp[2][i]) + sExp[3][i];
            if (v != 0.0) atomicAdd(dst + i, v);
        }
    }
    if (tid == 0) {
        nPairs[blockIdx.x] = myPairs;
        nTot[blockIdx.x] = myTot;
    }
}

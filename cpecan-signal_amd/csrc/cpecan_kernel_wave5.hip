/*
 * cpecan_kernel_wave5.hip -- the 5-state symbol machine (stateMachine5, impl/stateMachine.c:829-865; DNA against
 * DNA) with ONE WAVE per alignment and the recurrence in registers: posterior decode for bands up to 64 * L cells
 * (L = 1, 2, 3 cells per lane; cPecan's DNA bands with the default expansion are 41-120 cells wide).
 *
 * cpecan_kernel_general5.hip (one 256-thread workgroup per alignment) reads its two previous forward diagonals and
 * its backward diagonals back from memory behind a workgroup barrier on every diagonal; here nothing of the
 * recurrence leaves the wave:
 *   - slots are anchored to the matrix column: slot = x mod (64 * L), lane = slot / L.  Of a cell (x, y) on
 *     diagonal d = x + y the upper neighbour (x, y-1) is the slot's own cell on d-1, the lower (x-1, y) and middle
 *     (x-1, y-1) neighbours are the cells of slot - 1 on d-1 and d-2: a lane keeps its slots' five forward values of
 *     d-1, and one copy each of its lower neighbour's on d-1 and d-2 (the first becomes the second a diagonal later),
 *     so a diagonal costs ONE lane shift of five values.  Stale values of a slot that changed hands are never read:
 *     every neighbour is guarded by the band of its own diagonal, as the reference's NULL cells are;
 *   - forward cells are still written to the general kernel's [cell][state] layout in HBM: the sweep back needs
 *     them for the totals and the posteriors (loads that nothing waits on);
 *   - the traceback window is swept back in the same kernel (getPosteriorProbsWithBanding :921-992), the backward
 *     cells of d+1 and d+2 in registers the same way (neighbours at slot + 1); the order-dependent folds of
 *     diagonalCalculationTotalProbability (:736-754) and the ordered emission of the pairs (:756-795) go through an
 *     LDS staging row in x order, exactly as the general kernel's wave 0 does them.
 * Arithmetic, its order and the outputs are those of cpecan_kernel_general5.hip (and of the reference): the tests
 * compare both with the oracle bit for bit.  Expectations, un-banded calls, cell dumps and wider bands stay on the
 * general kernel.
 */
#include "cpecan_device.h"

enum {
    W5_MATCH_CONTINUE = 0, W5_MATCH_FROM_SHORT_GAP_X, W5_MATCH_FROM_LONG_GAP_X, W5_GAP_SHORT_OPEN_X,
    W5_GAP_SHORT_EXTEND_X, W5_GAP_SHORT_SWITCH_TO_X, W5_GAP_LONG_OPEN_X, W5_GAP_LONG_EXTEND_X,
    W5_GAP_LONG_SWITCH_TO_X, W5_MATCH_FROM_SHORT_GAP_Y, W5_MATCH_FROM_LONG_GAP_Y,
    W5_GAP_SHORT_OPEN_Y, W5_GAP_SHORT_EXTEND_Y, W5_GAP_SHORT_SWITCH_TO_Y, W5_GAP_LONG_OPEN_Y,
    W5_GAP_LONG_EXTEND_Y, W5_GAP_LONG_SWITCH_TO_Y
};
#define W5S 5

namespace {

__device__ __forceinline__ int w5_base(const char *s, long long i) {
    if (i < 0) return 4;
    const char ch = s[i];
    return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4;
}
__device__ __forceinline__ double w5_gap(const double *g, int i) { return i < 4 ? g[i] : CP_NEG_INF; }
__device__ __forceinline__ double w5_match(const double *m, int ix, int iy) {
    return ix < 4 && iy < 4 ? m[ix * 4 + iy] : CP_NEG_INF;
}
/* logAdd (impl/pairwiseAligner.c:251-255) and its lookup (:238-249) without branches, cp_logAdd's value bit for bit (a
 * wave that walks three cells per lane cannot afford cp_logAdd's divergent paths): the cubic's four float-literal
 * coefficients come from a 16-entry LDS table indexed by n = ceil(2d) -- the pieces' limits 1, 2.5 and 4.5 are
 * multiples of 1/2 and 2d is exact, so n decides the piece without a comparison -- and the reference's two early exits
 * ("the smaller operand is -inf", "d >= 7.5") are the one test d < 7.5: +inf and NaN fail it too. */
__device__ __forceinline__ double w5_ladd_t(double x, double y, const double *coef) {
    const bool xs = x < y;
    const double hi = xs ? y : x, lo = xs ? x : y;
    const double d = hi - lo;
    const bool near = d < 7.5;
    const int n = near ? (int) __builtin_ceil(d + d) : 15;
    const double *c = coef + 4 * n;
    const double r = (((c[0] * d + c[1]) * d + c[2]) * d + c[3]) + lo;
    return near ? r : hi;
}
__device__ __forceinline__ void w5_init_coef(double *coef) {
    const float t[16] = { -0.009350833524763f, 0.130659527668286f, 0.498799810682272f, 0.693203116424741f,
                          -0.014532321752540f, 0.139942324101744f, 0.495635523139337f, 0.692140569840976f,
                          -0.004605031767994f, 0.063427417320019f, 0.695956496475118f, 0.514272634594009f,
                          -0.000458661602210f, 0.009695946122598f, 0.930734667215156f, 0.168037164329057f };
    const int l = threadIdx.x & 63, n = l >> 2, piece = n <= 2 ? 0 : n <= 5 ? 1 : n <= 9 ? 2 : 3;
    coef[l] = (double) t[piece * 4 + (l & 3)];
}
#define cp_logAdd(a, b) w5_ladd_t((a), (b), coef) /* (everything below, where `coef` is the LDS table; the sequential
                                                       folds of cpecan_device.h keep their own) */
__device__ __forceinline__ double w5_match_from(const double *middle, double eP, const double *t, const double *coef) {
    double m = CP_NEG_INF;
    m = cp_logAdd(m, middle[0] + (eP + t[W5_MATCH_CONTINUE]));
    m = cp_logAdd(m, middle[1] + (eP + t[W5_MATCH_FROM_SHORT_GAP_X]));
    m = cp_logAdd(m, middle[2] + (eP + t[W5_MATCH_FROM_SHORT_GAP_Y]));
    m = cp_logAdd(m, middle[3] + (eP + t[W5_MATCH_FROM_LONG_GAP_X]));
    m = cp_logAdd(m, middle[4] + (eP + t[W5_MATCH_FROM_LONG_GAP_Y]));
    return m;
}
/* a wave runs in lockstep and its LDS operations execute in order: a lane's store is seen by another lane's later load
 * of the same wave.  All that is needed is that the compiler keeps the order (a fence at WAVEFRONT scope: no
 * instruction; a workgroup-scope fence would wait for every outstanding global load and store -- the prefetches -- on
 * every diagonal) */
__device__ __forceinline__ void w5_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
/* the matrix column slot `sl` holds on a diagonal whose band is [xmin, xmax] (at most P columns): the one congruent
 * to sl modulo P, or -1 */
template <int P> __device__ __forceinline__ int w5_column(int sl, int xmin, int xmax) {
    int x = xmin + ((sl - xmin) % P + P) % P;
    return x <= xmax ? x : -1;
}

/* PAIRED: the alignment has a workgroup of TWO waves -- wave 0 sweeps forward and publishes, at every traceback point,
 * the diagonal it has reached (its cells are in HBM by then: release fence, then the LDS word); wave 1 finds the same
 * traceback points from the band table alone, waits for the forward sweep to reach each of them and sweeps the window
 * back.  Same arithmetic, same order, same outputs as the one-wave form: the sweep back of a window never depended on
 * anything of the forward sweep but its cells up to the window's top.  A batch of fewer alignments than the chip has
 * SIMDs to spare gets two waves per SIMD this way (forward and backward sweeps overlap), and each wave holds only its
 * own half of the register state. */
/* ROLE 0: one wave does both sweeps; 1: the forward wave of a pair; 2: the backward wave of a pair (compile-time, so
 * that each wave of a pair is allocated the registers of its own sweep only) */
template <int L, bool EM, int ROLE> __device__ __forceinline__ void wave5_body(
    const DevItem &it, const DevParams &P, const int *__restrict__ bandL, const int *__restrict__ bandR,
    const long long *__restrict__ pre, const char *__restrict__ cx, const char *__restrict__ cy,
    const double *__restrict__ model, double *F, long long *pairs, double *pairLogp, long long *totXay,
    double *totVal, long long &myPairs, long long &myTot, double *stage /* LDS, 64 * L + 24 doubles */,
    const double *coef /* LDS, the logAdd table */, double *emis /* LDS, EM: the 80 emission sums */,
    double *expect /* EM: this model's [25 transitions | 80 emissions | likelihood] */,
    volatile long long *progress /* LDS, PAIRED: the last traceback point the forward sweep has reached */) {
    constexpr int PP = 64 * L;
    constexpr bool PAIRED = ROLE != 0, sweepsForward = ROLE != 2, sweepsBack = ROLE != 1;
    /* EM (diagonalCalculation_Expectations :841-863 over stateMachine5_cellCalculate with cell_updateExpectations
     * :407-424): a lane sums the thirteen kinds of transition over its cells in registers; the emission counts
     * [to-state][x base][y base] go to the LDS table; the likelihood is added once per diagonal (quirk Q7) */
    double trans[13];
#pragma unroll
    for (int k = 0; k < 13; k++) trans[k] = 0.0;
    double lik = 0.0;
    const int lane = threadIdx.x & 63;
    /* the emission tables in LDS (a cell's lookups are on its critical path), the transitions uniform */
    if (lane < 24) stage[PP + lane] = model[24 + lane];
    w5_wave_sync();
    const double *t = model, *mm = stage + PP, *gx = stage + PP + 16, *gy = stage + PP + 20;
    const long long D = it.lX + it.lY;

    /* forward state per slot: own cell on d-1, the lower neighbour's cells on d-1 and d-2 */
    double U[L][W5S], N1[L][W5S], N2[L][W5S];
#pragma unroll
    for (int j = 0; j < L; j++)
#pragma unroll
        for (int s = 0; s < W5S; s++) U[j][s] = N1[j][s] = N2[j][s] = CP_NEG_INF;
    /* diagonal 0: stateMachine5_startStateProb / raggedStartStateProb (:743-763); its only cell is column 0 = slot 0 */
    if (lane == 0 && sweepsForward) {
        U[0][0] = it.raggedL ? CP_NEG_INF : 0.0;
        U[0][1] = CP_NEG_INF;
        U[0][2] = CP_NEG_INF;
        U[0][3] = it.raggedL ? 0.0 : CP_NEG_INF;
        U[0][4] = it.raggedL ? 0.0 : CP_NEG_INF;
#pragma unroll
        for (int s = 0; s < W5S; s++) F[s] = U[0][s];
    }

    long long tracedBackTo = 0;
    int xmin1 = 0, xmax1 = 0;   /* band of d-1 (diagonal 0: the single column 0) */
    int xmin2 = 0, xmax2 = -1;  /* band of d-2 (none yet) */
    /* Nothing the sweep loads may sit on its critical path (one wave per SIMD: nobody else to run meanwhile): the
     * band row and the cell prefix of the next diagonal, and every slot's next y base, are fetched a diagonal ahead;
     * a slot's x base only changes when the slot changes hands. */
    int nLo = bandL[1], nHi = bandR[1];
    long long nPre = pre[1];
    int colX[L], baseX[L], nextBy[L];
#pragma unroll
    for (int j = 0; j < L; j++) {
        colX[j] = -2;
        baseX[j] = 4;
        const int x = w5_column<PP>(lane * L + j, (int) ((1 + nLo) / 2), (int) ((1 + nHi) / 2));
        nextBy[j] = x >= 0 ? w5_base(cy, (1 - (long long) x) - 1) : 4;
    }
    for (long long d = 1; d <= D; d++) {
        int lo, hi, xmin, xmax, width;
        long long preD;
        if (PAIRED && !sweepsForward) {
            /* the next traceback point at or after d: the last diagonal, or the first one minDiags past the previous
             * window whose band is narrow enough (the test of the one-wave loop below), 64 diagonals per step */
            long long from = tracedBackTo + P.minDiags;
            if (from < d) from = d;
            long long dTopNext = D;
            for (long long base = from; base < D; base += 64) {
                const long long dd = base + lane;
                bool narrow = false;
                if (dd < D) {
                    const int l = bandL[dd], h = bandR[dd];
                    narrow = (int) ((dd + h) / 2) - (int) ((dd + l) / 2) + 1 <= P.expansion * 2 + 1;
                }
                const unsigned long long m = __ballot(narrow);
                if (m) {
                    dTopNext = base + (__ffsll((long long) m) - 1);
                    break;
                }
            }
            d = dTopNext;
            lo = bandL[d];
            hi = bandR[d];
            preD = pre[d];
            xmin = (int) ((d + lo) / 2);
            xmax = (int) ((d + hi) / 2);
            width = xmax - xmin + 1;
        } else {
        lo = nLo;
        hi = nHi;
        preD = nPre;
        xmin = (int) ((d + lo) / 2);
        xmax = (int) ((d + hi) / 2);
        width = xmax - xmin + 1;
        if (d < D) {
            nLo = bandL[d + 1];
            nHi = bandR[d + 1];
            nPre = pre[d + 1];
        }
        int by[L];
#pragma unroll
        for (int j = 0; j < L; j++) by[j] = nextBy[j];
        if (d < D) {
            const int nxmin = (int) ((d + 1 + nLo) / 2), nxmax = (int) ((d + 1 + nHi) / 2);
#pragma unroll
            for (int j = 0; j < L; j++) {
                const int x = w5_column<PP>(lane * L + j, nxmin, nxmax);
                nextBy[j] = x >= 0 ? w5_base(cy, (d + 1 - (long long) x) - 1) : 4;
            }
        }
        /* the lower neighbour's cells of d-1: slot - 1 is the previous layer of the lane, or the last layer of the
         * lane below */
#pragma unroll
        for (int s = 0; s < W5S; s++) {
            const double below = __shfl(U[L - 1][s], (lane + 63) & 63);
#pragma unroll
            for (int j = L - 1; j >= 1; j--) {
                N2[j][s] = N1[j][s];
                N1[j][s] = U[j - 1][s];
            }
            N2[0][s] = N1[0][s];
            N1[0][s] = below;
        }
        double *fd = F + preD * W5S;
#pragma unroll
        for (int j = 0; j < L; j++) {
            const int sl = lane * L + j;
            const int x = w5_column<PP>(sl, xmin, xmax);
            double o[W5S];
#pragma unroll
            for (int s = 0; s < W5S; s++) o[s] = CP_NEG_INF;
            if (x >= 0) {
                if (x != colX[j]) {
                    colX[j] = x;
                    baseX[j] = w5_base(cx, (long long) x - 1);
                }
                const int bx = baseX[j];
                const bool lower = x - 1 >= xmin1 && x - 1 <= xmax1;
                const bool middle = d >= 2 && x - 1 >= xmin2 && x - 1 <= xmax2;
                const bool upper = x >= xmin1 && x <= xmax1;
                if (lower) {
                    const double eP = w5_gap(gx, bx);
                    o[1] = cp_logAdd(o[1], N1[j][0] + (eP + t[W5_GAP_SHORT_OPEN_X]));
                    o[1] = cp_logAdd(o[1], N1[j][1] + (eP + t[W5_GAP_SHORT_EXTEND_X]));
                    o[3] = cp_logAdd(o[3], N1[j][0] + (eP + t[W5_GAP_LONG_OPEN_X]));
                    o[3] = cp_logAdd(o[3], N1[j][3] + (eP + t[W5_GAP_LONG_EXTEND_X]));
                }
                if (middle) o[0] = w5_match_from(N2[j], w5_match(mm, bx, by[j]), t, coef);
                if (upper) {
                    const double eP = w5_gap(gy, by[j]);
                    o[2] = cp_logAdd(o[2], U[j][0] + (eP + t[W5_GAP_SHORT_OPEN_Y]));
                    o[2] = cp_logAdd(o[2], U[j][2] + (eP + t[W5_GAP_SHORT_EXTEND_Y]));
                    o[4] = cp_logAdd(o[4], U[j][0] + (eP + t[W5_GAP_LONG_OPEN_Y]));
                    o[4] = cp_logAdd(o[4], U[j][4] + (eP + t[W5_GAP_LONG_EXTEND_Y]));
                }
                double *dst = fd + (long long) (x - xmin) * W5S;
#pragma unroll
                for (int s = 0; s < W5S; s++) dst[s] = o[s];
            }
#pragma unroll
            for (int s = 0; s < W5S; s++) U[j][s] = o[s];
        }
        xmin2 = xmin1; xmax2 = xmax1;
        xmin1 = xmin; xmax1 = xmax;
        }

        const bool atEnd = d == D;
        const bool tb = d >= tracedBackTo + P.minDiags && width <= P.expansion * 2 + 1;
        if (!(atEnd || tb)) continue;

        /* ---- traceback window (:921-992) ---- */
        const long long dTop = d;
        const long long tracedBackFrom = dTop - (atEnd ? 0 : P.tbDiags + 1);
        if (PAIRED) {
            if (sweepsForward) { /* every cell up to dTop is written: say so, and go on */
                __threadfence();
                if (lane == 0) *progress = dTop;
                tracedBackTo = tracedBackFrom;
                continue;
            }
            while (*progress < dTop) __builtin_amdgcn_s_sleep(16);
            __threadfence(); /* (acquire: nothing of this CU's vector cache predates the forward wave's stores) */
        } else
            __threadfence_block(); /* the window's forward cells are read back from HBM below */
        double e[W5S]; /* stateMachine5_endStateProb / raggedEndStateProb (:765-789) */
        if (atEnd && it.raggedR) {
            e[0] = t[W5_GAP_LONG_OPEN_X];
            e[1] = t[W5_GAP_LONG_OPEN_X];
            e[2] = t[W5_GAP_LONG_OPEN_Y];
            e[3] = t[W5_GAP_LONG_EXTEND_X];
            e[4] = t[W5_GAP_LONG_EXTEND_Y];
        } else {
            e[0] = t[W5_MATCH_CONTINUE];
            e[1] = t[W5_MATCH_FROM_SHORT_GAP_X];
            e[2] = t[W5_MATCH_FROM_SHORT_GAP_Y];
            e[3] = t[W5_MATCH_FROM_LONG_GAP_X];
            e[4] = t[W5_MATCH_FROM_LONG_GAP_Y];
        }
        /* backward state per slot: own cell on d2+1 (Bo), the upper-slot neighbour's cells on d2+1 and d2+2 */
        double Bo[L][W5S], Bn1[L][W5S], Bn2[L][W5S];
#pragma unroll
        for (int j = 0; j < L; j++)
#pragma unroll
            for (int s = 0; s < W5S; s++) Bo[j][s] = Bn1[j][s] = Bn2[j][s] = CP_NEG_INF;
        int bxmin1 = 0, bxmax1 = -1, bxmin2 = 0, bxmax2 = -1; /* bands of d2+1 and d2+2 (none above dTop) */
        double total = CP_NEG_INF;
        long long calcs = 0;
        /* as on the way up, what the sweep loads is fetched a diagonal ahead: the band row and cell prefix, every
         * slot's y base and the match-state forward value of its cell (the other four states are only read on the
         * one diagonal in ten that refreshes the total) */
        int qLo = lo, qHi = hi;
        long long qPre = preD;
        int colB[L], baseXb[L], nextYb[L];
        double nextF0[L];
#pragma unroll
        for (int j = 0; j < L; j++) {
            colB[j] = -2;
            baseXb[j] = 4;
            const int x = w5_column<PP>(lane * L + j, xmin, xmax);
            nextYb[j] = x >= 0 ? w5_base(cy, dTop - (long long) x) : 4;
            if (PAIRED)
                nextF0[j] = x >= 0 ? F[(preD + (x - xmin)) * W5S] : 0.0;
            else
                nextF0[j] = U[j][0]; /* the forward cells of dTop are still in registers */
        }
        for (long long d2 = dTop; d2 > tracedBackTo; d2--) {
            const int l2 = qLo, h2 = qHi;
            const long long pre2 = qPre;
            const int cxmin = (int) ((d2 + l2) / 2), cxmax = (int) ((d2 + h2) / 2), w2 = cxmax - cxmin + 1;
            int yb[L];
            double f0[L];
#pragma unroll
            for (int j = 0; j < L; j++) {
                yb[j] = nextYb[j];
                f0[j] = nextF0[j];
            }
            if (d2 - 1 > tracedBackTo) {
                qLo = bandL[d2 - 1];
                qHi = bandR[d2 - 1];
                qPre = pre[d2 - 1];
                const int nxmin = (int) ((d2 - 1 + qLo) / 2), nxmax = (int) ((d2 - 1 + qHi) / 2);
#pragma unroll
                for (int j = 0; j < L; j++) {
                    const int x = w5_column<PP>(lane * L + j, nxmin, nxmax);
                    nextYb[j] = x >= 0 ? w5_base(cy, d2 - 1 - (long long) x) : 4;
                    nextF0[j] = (x >= 0 && d2 - 1 <= tracedBackFrom) ? F[(qPre + (x - nxmin)) * W5S] : 0.0;
                }
            }
            double cur[L][W5S];
#pragma unroll
            for (int j = 0; j < L; j++) {
                const int sl = lane * L + j;
                const int x = w5_column<PP>(sl, cxmin, cxmax);
                if (x >= 0 && x != colB[j]) {
                    colB[j] = x;
                    baseXb[j] = w5_base(cx, x);
                }
#pragma unroll
                for (int s = 0; s < W5S; s++) cur[j][s] = d2 == dTop ? e[s] : CP_NEG_INF;
                if (x >= 0 && d2 < dTop) {
                    /* (ii) cell (x+1, y+1) on d2+2 reaches this cell through its middle block */
                    if (d2 + 2 <= dTop && x + 1 >= bxmin2 && x + 1 <= bxmax2) {
                        const double eP = w5_match(mm, baseXb[j], yb[j]);
                        const double s20 = Bn2[j][0];
                        cur[j][0] = cp_logAdd(cur[j][0], s20 + (eP + t[W5_MATCH_CONTINUE]));
                        cur[j][1] = cp_logAdd(cur[j][1], s20 + (eP + t[W5_MATCH_FROM_SHORT_GAP_X]));
                        cur[j][2] = cp_logAdd(cur[j][2], s20 + (eP + t[W5_MATCH_FROM_SHORT_GAP_Y]));
                        cur[j][3] = cp_logAdd(cur[j][3], s20 + (eP + t[W5_MATCH_FROM_LONG_GAP_X]));
                        cur[j][4] = cp_logAdd(cur[j][4], s20 + (eP + t[W5_MATCH_FROM_LONG_GAP_Y]));
                    }
                    /* (iii) cell (x, y+1) on d2+1 reaches it through its upper block */
                    if (x >= bxmin1 && x <= bxmax1) {
                        const double eP = w5_gap(gy, yb[j]);
                        cur[j][0] = cp_logAdd(cur[j][0], Bo[j][2] + (eP + t[W5_GAP_SHORT_OPEN_Y]));
                        cur[j][2] = cp_logAdd(cur[j][2], Bo[j][2] + (eP + t[W5_GAP_SHORT_EXTEND_Y]));
                        cur[j][0] = cp_logAdd(cur[j][0], Bo[j][4] + (eP + t[W5_GAP_LONG_OPEN_Y]));
                        cur[j][4] = cp_logAdd(cur[j][4], Bo[j][4] + (eP + t[W5_GAP_LONG_EXTEND_Y]));
                    }
                    /* (iv) cell (x+1, y) on d2+1 reaches it through its lower block */
                    if (x + 1 >= bxmin1 && x + 1 <= bxmax1) {
                        const double eP = w5_gap(gx, baseXb[j]);
                        cur[j][0] = cp_logAdd(cur[j][0], Bn1[j][1] + (eP + t[W5_GAP_SHORT_OPEN_X]));
                        cur[j][1] = cp_logAdd(cur[j][1], Bn1[j][1] + (eP + t[W5_GAP_SHORT_EXTEND_X]));
                        cur[j][0] = cp_logAdd(cur[j][0], Bn1[j][3] + (eP + t[W5_GAP_LONG_OPEN_X]));
                        cur[j][3] = cp_logAdd(cur[j][3], Bn1[j][3] + (eP + t[W5_GAP_LONG_EXTEND_X]));
                    }
                }
            }

            if (d2 <= tracedBackFrom) {
                const double *fdd = F + pre2 * W5S;
                if (calcs++ % 10 == 0) {
                    /* diagonalCalculationTotalProbability :736-754: the cells' dot products folded in x order */
                    w5_wave_sync();
#pragma unroll
                    for (int j = 0; j < L; j++) {
                        const int x = w5_column<PP>(lane * L + j, cxmin, cxmax);
                        if (x >= 0) {
                            const double *f = fdd + (long long) (x - cxmin) * W5S;
                            double v = f[0] + cur[j][0]; /* cell_dotProduct :391-397 */
#pragma unroll
                            for (int s = 1; s < W5S; s++) v = cp_logAdd(v, f[s] + cur[j][s]);
                            stage[x - cxmin] = v;
                        }
                    }
                    w5_wave_sync();
                    double acc = CP_NEG_INF;
                    for (int base = 0; base < w2; base += 64) {
                        const int cc = base + lane;
                        acc = cp_wave_seq_fold(acc, cc < w2 ? stage[cc] : CP_NEG_INF, cc < w2);
                    }
                    if (d2 + 1 <= dTop) {
                        /* matches that step over d2: forward[d2-1] --match--> cells of d2+1 (whose backward values
                         * are the own-slot registers Bo) */
                        const int w3 = bxmax1 - bxmin1 + 1;
                        w5_wave_sync();
#pragma unroll
                        for (int j = 0; j < L; j++) {
                            const int x = w5_column<PP>(lane * L + j, bxmin1, bxmax1);
                            if (x >= 0) {
                                const long long y = d2 + 1 - x;
                                double m = CP_NEG_INF;
                                const long long dm = d2 - 1;
                                if (dm >= 0) {
                                    const int ml = bandL[dm], mh = bandR[dm];
                                    const int mxmin = (int) ((dm + ml) / 2), mxmax = (int) ((dm + mh) / 2);
                                    if (x - 1 >= mxmin && x - 1 <= mxmax) {
                                        const double *mid = F + (pre[dm] + (x - 1 - mxmin)) * W5S;
                                        m = w5_match_from(mid, w5_match(mm, w5_base(cx, (long long) x - 1), w5_base(cy, y - 1)), t, coef);
                                    }
                                }
                                double v = m + Bo[j][0];
#pragma unroll
                                for (int s = 1; s < W5S; s++) v = cp_logAdd(v, CP_NEG_INF + Bo[j][s]);
                                stage[x - bxmin1] = v;
                            }
                        }
                        w5_wave_sync();
                        double acc2 = CP_NEG_INF;
                        for (int base = 0; base < w3; base += 64) {
                            const int cc = base + lane;
                            acc2 = cp_wave_seq_fold(acc2, cc < w3 ? stage[cc] : CP_NEG_INF, cc < w3);
                        }
                        acc = cp_logAdd(acc, acc2);
                    }
                    if (lane == 0 && myTot < it.totCap) {
                        totXay[it.totBase + myTot] = d2;
                        totVal[it.totBase + myTot] = acc;
                    }
                    myTot++;
                    total = acc;
                }

                if (EM) {
                    if (lane == 0) lik += total;
                    const bool haveMiddle = d2 - 2 >= tracedBackTo; /* forward[d2-2] is freed otherwise (:982) */
                    int e1min = 0, e1max = -1, e2min = 0, e2max = -1;
                    long long e1pre = 0, e2pre = 0;
                    if (d2 >= 1) {
                        const long long dd = d2 - 1;
                        e1min = (int) ((dd + bandL[dd]) / 2);
                        e1max = (int) ((dd + bandR[dd]) / 2);
                        e1pre = pre[dd];
                    }
                    if (d2 >= 2 && haveMiddle) {
                        const long long dd = d2 - 2;
                        e2min = (int) ((dd + bandL[dd]) / 2);
                        e2max = (int) ((dd + bandR[dd]) / 2);
                        e2pre = pre[dd];
                    }
#pragma unroll
                    for (int j = 0; j < L; j++) {
                        const int x = w5_column<PP>(lane * L + j, cxmin, cxmax);
                        if (x < 0) continue;
                        const long long y = d2 - x;
                        const int bx = w5_base(cx, (long long) x - 1), by = w5_base(cy, y - 1);
                        double into[W5S] = { 0.0, 0.0, 0.0, 0.0, 0.0 };
                        if (x - 1 >= e1min && x - 1 <= e1max) { /* lower */
                            const double *nb = F + (e1pre + (x - 1 - e1min)) * W5S;
                            const double eP = w5_gap(gx, bx);
                            double pr;
                            pr = exp(nb[0] + cur[j][1] + (eP + t[W5_GAP_SHORT_OPEN_X]) - total); trans[0] += pr; into[1] += pr;
                            pr = exp(nb[1] + cur[j][1] + (eP + t[W5_GAP_SHORT_EXTEND_X]) - total); trans[1] += pr; into[1] += pr;
                            pr = exp(nb[0] + cur[j][3] + (eP + t[W5_GAP_LONG_OPEN_X]) - total); trans[2] += pr; into[3] += pr;
                            pr = exp(nb[3] + cur[j][3] + (eP + t[W5_GAP_LONG_EXTEND_X]) - total); trans[3] += pr; into[3] += pr;
                        }
                        if (haveMiddle && x - 1 >= e2min && x - 1 <= e2max) { /* middle */
                            const double *nb = F + (e2pre + (x - 1 - e2min)) * W5S;
                            const double eP = w5_match(mm, bx, by);
                            double pr;
                            pr = exp(nb[0] + cur[j][0] + (eP + t[W5_MATCH_CONTINUE]) - total); trans[4] += pr; into[0] += pr;
                            pr = exp(nb[1] + cur[j][0] + (eP + t[W5_MATCH_FROM_SHORT_GAP_X]) - total); trans[5] += pr; into[0] += pr;
                            pr = exp(nb[2] + cur[j][0] + (eP + t[W5_MATCH_FROM_SHORT_GAP_Y]) - total); trans[6] += pr; into[0] += pr;
                            pr = exp(nb[3] + cur[j][0] + (eP + t[W5_MATCH_FROM_LONG_GAP_X]) - total); trans[7] += pr; into[0] += pr;
                            pr = exp(nb[4] + cur[j][0] + (eP + t[W5_MATCH_FROM_LONG_GAP_Y]) - total); trans[8] += pr; into[0] += pr;
                        }
                        if (x >= e1min && x <= e1max) { /* upper */
                            const double *nb = F + (e1pre + (x - e1min)) * W5S;
                            const double eP = w5_gap(gy, by);
                            double pr;
                            pr = exp(nb[0] + cur[j][2] + (eP + t[W5_GAP_SHORT_OPEN_Y]) - total); trans[9] += pr; into[2] += pr;
                            pr = exp(nb[2] + cur[j][2] + (eP + t[W5_GAP_SHORT_EXTEND_Y]) - total); trans[10] += pr; into[2] += pr;
                            pr = exp(nb[0] + cur[j][4] + (eP + t[W5_GAP_LONG_OPEN_Y]) - total); trans[11] += pr; into[4] += pr;
                            pr = exp(nb[4] + cur[j][4] + (eP + t[W5_GAP_LONG_EXTEND_Y]) - total); trans[12] += pr; into[4] += pr;
                        }
                        if (bx < 4 && by < 4) {
#pragma unroll
                            for (int st = 0; st < W5S; st++)
                                if (into[st] != 0.0) atomicAdd(&emis[st * 16 + bx * 4 + by], into[st]);
                        }
                    }
                } else {
                /* diagonalCalculationPosteriorMatchProbs :756-795, emitted in x order */
                w5_wave_sync();
#pragma unroll
                for (int j = 0; j < L; j++) {
                    const int x = w5_column<PP>(lane * L + j, cxmin, cxmax);
                    if (x >= 0) stage[x - cxmin] = f0[j] + cur[j][0];
                }
                w5_wave_sync();
                for (int base = 0; base < w2; base += 64) {
                    const int cc = base + lane;
                    bool hit = false;
                    double ee = 0.0, p = 0.0;
                    long long x = 0, y = 0;
                    if (cc < w2) {
                        x = cxmin + cc;
                        y = d2 - x;
                        if (x > 0 && y > 0) {
                            ee = stage[cc] - total;
                            p = exp(ee);
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
                            pairLogp[it.pairBase + idx] = ee;
                        }
                    }
                    myPairs += __popcll(m);
                }
                }
            }

            /* one diagonal down: the upper-slot neighbour's d2+1 becomes its d2+2, this diagonal becomes d2+1 */
#pragma unroll
            for (int s = 0; s < W5S; s++) {
                const double above = __shfl(cur[0][s], (lane + 1) & 63);
#pragma unroll
                for (int j = 0; j < L; j++) {
                    Bn2[j][s] = Bn1[j][s];
                    Bn1[j][s] = j + 1 < L ? cur[j + 1 < L ? j + 1 : j][s] : above;
                    Bo[j][s] = cur[j][s];
                }
            }
            bxmin2 = bxmin1; bxmax2 = bxmax1;
            bxmin1 = cxmin; bxmax1 = cxmax;
        }
        tracedBackTo = tracedBackFrom;
    }
    if (EM && sweepsBack) {
        /* [from * 5 + to] of the thirteen kinds, in the order they were summed above */
        const int slot[13] = { 0 * 5 + 1, 1 * 5 + 1, 0 * 5 + 3, 3 * 5 + 3, 0 * 5 + 0, 1 * 5 + 0, 2 * 5 + 0, 3 * 5 + 0, 4 * 5 + 0,
                               0 * 5 + 2, 2 * 5 + 2, 0 * 5 + 4, 4 * 5 + 4 };
#pragma unroll
        for (int k = 0; k < 13; k++) {
            double v = trans[k];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
            if (lane == 0 && v != 0.0) atomicAdd(expect + slot[k], v);
        }
        w5_wave_sync();
        for (int i = lane; i < 80; i += 64)
            if (emis[i] != 0.0) atomicAdd(expect + 25 + i, emis[i]);
        if (lane == 0) atomicAdd(expect + CP_EXPECT5_LEN - 1, lik);
    }
}

} // namespace

/* model block: [17 transitions | pad to 24 | 16 match | 4 gapX | 4 gapY] = CP_MODEL5_STRIDE doubles */
#define W5_OCC(PAIRED) W5_OCC_##PAIRED
#define W5_OCC_false
#define W5_OCC_true __attribute__((amdgpu_waves_per_eu(2))) /* two waves per SIMD at least: 256 registers */
#define W5_ARGS(EM)                                                                                               \
    (it, P, bandL + it.diagBase, bandR + it.diagBase, cellPrefix + it.diagBase, xChars + it.xOff, yChars + it.yOff, \
     models + (long long) it.model * CP_MODEL5_STRIDE, Fstore + it.cellBase * W5S, pairs, pairLogp, totXay, totVal, \
     myPairs, myTot, stage, coefTable, emis, EM ? expect + (long long) it.model * CP_EXPECT5_LEN : nullptr, &progress)
#define W5_CALL(L, EM, PAIRED)                                                                                    \
    {                                                                                                             \
        if (!PAIRED) wave5_body<L, EM, 0> W5_ARGS(EM);                                                            \
        else if (__builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6) == 0) wave5_body<L, EM, 1> W5_ARGS(EM);  \
        else wave5_body<L, EM, 2> W5_ARGS(EM);                                                                    \
    }
#define W5_KERNEL(L, NAME, EM, PAIRED)                                                                            \
    extern "C" __global__ __launch_bounds__(PAIRED ? 128 : 64) W5_OCC(PAIRED) void NAME(                                         \
        const DevItem *items, DevParams P, const int *bandL, const int *bandR, const long long *cellPrefix,       \
        const char *xChars, const char *yChars, const double *models, double *Fstore, long long *pairs,           \
        double *pairLogp, long long *nPairs, long long *totXay, double *totVal, long long *nTot, double *expect) { \
        __shared__ double stage[64 * L + 24];                                                                     \
        __shared__ double coefTable[64];                                                                          \
        __shared__ double emis[80];                                                                               \
        __shared__ long long progress;                                                                            \
        w5_init_coef(coefTable);                                                                                  \
        for (int i = threadIdx.x; i < 80; i += (PAIRED ? 128 : 64)) emis[i] = 0.0;                                \
        if (threadIdx.x == 0) progress = 0;                                                                       \
        if (PAIRED) __syncthreads();                                                                              \
        else w5_wave_sync();                                                                                      \
        const DevItem it = items[blockIdx.x];                                                                     \
        long long myPairs = 0, myTot = 0;                                                                         \
        if (it.lX + it.lY > 0)                                                                                    \
            W5_CALL(L, EM, PAIRED)                                                                                \
        if (threadIdx.x == (PAIRED ? 64 : 0)) {                                                                   \
            nPairs[blockIdx.x] = myPairs;                                                                         \
            nTot[blockIdx.x] = myTot;                                                                             \
        }                                                                                                         \
    }
W5_KERNEL(1, cpecan_k_wave5_l1, false, false)
W5_KERNEL(2, cpecan_k_wave5_l2, false, false)
W5_KERNEL(3, cpecan_k_wave5_l3, false, false)
W5_KERNEL(1, cpecan_k_wave5e_l1, true, false)
W5_KERNEL(2, cpecan_k_wave5e_l2, true, false)
W5_KERNEL(3, cpecan_k_wave5e_l3, true, false)
W5_KERNEL(1, cpecan_k_wave5p_l1, false, true)
W5_KERNEL(2, cpecan_k_wave5p_l2, false, true)
W5_KERNEL(3, cpecan_k_wave5p_l3, false, true)
W5_KERNEL(1, cpecan_k_wave5pe_l1, true, true)
W5_KERNEL(2, cpecan_k_wave5pe_l2, true, true)
W5_KERNEL(3, cpecan_k_wave5pe_l3, true, true)

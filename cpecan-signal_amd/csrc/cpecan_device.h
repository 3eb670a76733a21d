/*
 * cpecan_device.h -- device-side building blocks shared by the gfx950 kernels.
 *
 * Arithmetic contract (what makes the kernels bit-identical to the reference's C code and to the
 * CPU oracle on the same inputs): fp64 throughout, this translation unit is built with
 * -ffp-contract=off so every multiply and add rounds separately like the reference's x86-64 build,
 * the float-suffixed literals of lookup() are kept as floats, and the accumulation order of every
 * logAdd chain follows the reference (impl/pairwiseAligner.c, impl/stateMachine.c; cited inline).
 */
#ifndef CPECAN_DEVICE_H_
#define CPECAN_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CP_NEG_INF (-__builtin_huge_val())

/* derived model table: one row of CP_ROW doubles per k-mer index 0..4096 (4096 = "not a k-mer") */
#define CP_ROW 18
#define CP_MU 0
#define CP_SD 1
#define CP_RSD 2
#define CP_K1 3
#define CP_NMU 4
#define CP_NSD 5
#define CP_RNSD 6
#define CP_K2 7
#define CP_YMU 8
#define CP_YSD 9
#define CP_RYSD 10
#define CP_YK1 11
#define CP_YNMU 12
#define CP_YNSD 13
#define CP_RYNSD 14
#define CP_YK2 15
#define CP_GAPX 16
#define CP_MODEL5_STRIDE 48 /* 5-state symbol model: 17 transitions, pad to 24, 16 match, 4 gapX, 4 gapY */
#define CP_EXPECTV_LEN 61   /* vanilla machine's Baum-Welch sums: 30 beta + 30 alpha skip bins, likelihood */
#define CP_EXPECT5_LEN 106  /* its Baum-Welch sums: 25 transitions, 5 x 16 emissions, likelihood */
/* vanilla signal model block: header (scalars, per-bin log transition probabilities), then one row per
 * k-mer (4096 + the "not a k-mer" row): the match table's six values, then the extra-event table's */
#define CP_VHDR 160
#define CP_VHDR_END_M 2
#define CP_VHDR_END_X 3
#define CP_VHDR_END_Y 4
#define CP_VHDR_LOG_YY 5
#define CP_VHDR_LOG_YM 6
#define CP_VHDR_BINS 8 /* 30 x [log a_mx, log a_xx, log a_mm, log a_xm, log a_my] */
#define CP_VROW 12
#define CP_V_MU 0
#define CP_V_SD 1
#define CP_V_K 2
#define CP_V_NMU 3
#define CP_V_LAMBDA 4
#define CP_V_LLAMBDA 5
#define CP_VMODEL_STRIDE (CP_VHDR + 4097 * CP_VROW)
#define CP_MODEL_HEADER 16 /* doubles in front of the rows: the 9 transitions */
#define CP_MODEL_STRIDE (CP_MODEL_HEADER + 4097 * CP_ROW)

/* transition slots, order of struct _StateMachine3 (inc/stateMachine.h:179-187) */
#define T_MATCH_CONTINUE 0
#define T_MATCH_FROM_GAP_X 1
#define T_MATCH_FROM_GAP_Y 2
#define T_GAP_OPEN_X 3
#define T_GAP_OPEN_Y 4
#define T_GAP_EXTEND_X 5
#define T_GAP_EXTEND_Y 6
#define T_GAP_SWITCH_TO_X 7
#define T_GAP_SWITCH_TO_Y 8

struct DevItem {
    long long lX, lY;
    long long xOff, yOff;     /* into kidx[] / events[] (in events) */
    long long anchorOff, nAnchors;
    long long diagBase;       /* into bandL/bandR/cellPrefix */
    long long cellBase;       /* into the forward-cell store (cells) */
    long long nCells;
    long long pairBase, pairCap;
    long long totBase, totCap;
    long long bwsBase;        /* into the backward workspace (doubles) */
    int model, raggedL, raggedR, maxWidth;
};

struct DevParams {
    double threshold;
    long long minDiags, tbDiags, expansion;
    int mode, debug, unbanded;
    int scanDecode; /* systolic kernels: decode posteriors by the full scan (diagnostic) */
    double logThrSlack; /* log(threshold) minus a safety margin: cells below it skip exp() */
    int ldsWidth; /* 5-state general kernel: cells per diagonal the workgroup's LDS holds -- the forward sweep reads its
                     two previous diagonals from there instead of HBM (0: band too wide, through HBM).  The backward
                     diagonals stay in HBM: with them in LDS too a CU holds three workgroups instead of five and the
                     batch runs slower (4.1 against 6.3 Gcells/s) */
};

/* lookup(): impl/pairwiseAligner.c:238-249 -- four cubics, float literals */
__device__ __forceinline__ double cp_lookup(double x) {
    if (x <= 1.00f)
        return ((-0.009350833524763f * x + 0.130659527668286f) * x + 0.498799810682272f) * x
               + 0.693203116424741f;
    if (x <= 2.50f)
        return ((-0.014532321752540f * x + 0.139942324101744f) * x + 0.495635523139337f) * x
               + 0.692140569840976f;
    if (x <= 4.50f)
        return ((-0.004605031767994f * x + 0.063427417320019f) * x + 0.695956496475118f) * x
               + 0.514272634594009f;
    return ((-0.000458661602210f * x + 0.009695946122598f) * x + 0.930734667215156f) * x
           + 0.168037164329057f;
}

/* logAdd(): impl/pairwiseAligner.c:251-255 */
__device__ __forceinline__ double cp_logAdd(double x, double y) {
    if (x < y) return (x == CP_NEG_INF || y - x >= 7.5) ? y : cp_lookup(y - x) + x;
    return (y == CP_NEG_INF || x - y >= 7.5) ? x : cp_lookup(x - y) + y;
}

/* emissions_signal_logGaussPdf impl/stateMachine.c:333-343 with the constant part
 * K = log_inv_sqrt_2pi - log(sigma) taken from the host-built table */
__device__ __forceinline__ double cp_logGauss(double x, double mu, double sd, double K) {
    if (sd == 0.0) return CP_NEG_INF;
    double a = (x - mu) / sd;
    return K + (-0.5 * a * a);
}

/* exact sequential logAdd fold over the lanes of one wave, in lane order:
 *     for lane in 0..63: if valid[lane]: acc = logAdd(acc, v[lane])
 * (the order-dependent fold of dpDiagonal_dotProduct, impl/pairwiseAligner.c:587-597).  A term
 * leaves acc unchanged exactly when it is -inf or lies >= 7.5 below acc, so the loop only visits
 * the lanes that change the running value; acc is wave-uniform on entry and exit. */
__device__ __forceinline__ double cp_wave_seq_fold(double acc, double v, bool valid) {
    const int lane = threadIdx.x & 63;
    unsigned long long after = ~0ull; /* lanes still to be visited */
    for (;;) {
        bool eff = valid && ((after >> lane) & 1ull) && (v > CP_NEG_INF) && !(acc - v >= 7.5);
        unsigned long long m = __ballot(eff);
        if (m == 0ull) break;
        int first = __ffsll((long long) m) - 1;
        double vv = __shfl(v, first);
        acc = cp_logAdd(acc, vv);
        after = first >= 63 ? 0ull : (~0ull << (first + 1));
    }
    return acc;
}

/* one HDP model on the device (cpecan_hip_modelsh_create): the 3-state transitions and the NanoporeHDP as densities
 * need it */
struct DevHdpModel {
    double t[9];            /* transitions, order of struct _StateMachine3_HDP */
    int gridLength, pad;
    const int *kmerRow;     /* [alphabetSize^6] table row per k-mer id */
    const double *grid;     /* [gridLength] */
    const double *y;        /* [rows][gridLength] */
    const double *slope;    /* [rows][gridLength] */
};

#endif

/*
 * cpecan_asm.h -- what the C-ABI layer (cpecan_hip.hip) and the hand-scheduled assembly sweeps (asm/gen_sweeps.py ->
 * asm/cpecan_sweeps_gfx950.s) share: the kernel argument block, the host-built plan of an alignment's traceback windows
 * and band steps, and the launch entry points of cpecan_asm.hip.
 *
 * The assembly sweeps are the strawMan signal machine's forward and backward sweeps of cpecan_kernel_wave.hip (three
 * cells per lane, posterior decode, no gap-Y -> gap-X transition) with the instruction schedule written by hand; every
 * buffer they touch has the format of the compiled kernels, which remain the path of every other batch.
 */
#ifndef CPECAN_ASM_H_
#define CPECAN_ASM_H_

#include <hip/hip_runtime.h>

#include "cpecan_asm_gen.h"
#include "cpecan_device.h"

/* One traceback window of one alignment (getPosteriorProbsWithBanding's schedule, impl/pairwiseAligner.c:917-921, is a
 * function of the band alone: the host works it out with the band). */
struct AsmPlanWin {
    int d0;          /* last diagonal done before this window's forward launch (0 for the first) */
    int top;         /* the traceback point: last diagonal of the launch, first of the sweep back */
    int from, to;    /* tracedBackFrom, tracedBackTo */
    int atEnd;
    int xminTop, xmaxTop; /* band of `top` */
    int nWindows;    /* of this alignment (the same in all its records) */
    long long cells; /* cells of diagonals 0..top */
    int xmin0, xmax0; /* band of d0 */
    int tpost0;      /* first decoded diagonal: min(top, from) */
    int pad[3];
};
static_assert(sizeof(AsmPlanWin) == ASM_PLANWIN_BYTES, "plan record");

/* Per block of ASM_BLOCK diagonals: bit d % 64 of stepMin / stepMax = the band's first / last column moved up by one
 * coming to diagonal d; of full = the forward sweep keeps all three states of diagonal d in the ring. */
struct AsmPlanCtl {
    unsigned long long stepMin, stepMax, full, spare;
};
static_assert(sizeof(AsmPlanCtl) == ASM_CTL_BYTES, "control block");

struct AsmArgs {
    const DevItem *items;
    const long long *trackBase;
    const AsmPlanWin *planWin;   /* [item][maxWindows] */
    const AsmPlanCtl *planCtl;
    const long long *planOff;    /* first control block of every item */
    const double *events;
    const double *models;
    const double *track;
    double *ring;
    long long ringDoubles;
    void *states;
    char *ctx;                   /* [item][3] contexts of ctxBytes */
    long long ctxBytes;
    const double *coef;          /* the logAdd table as init_coef() lays it out: 64 doubles */
    int nItems, window, ringD, maxWindows;
    char *scratch;
    long long scratchBytes;
    double logThrSlack;
    long long modelStride;       /* doubles */
    const unsigned *maskTab;     /* ASM_MASK_BYTES per diagonal of every item (cpecan_k_asm_masks), at the item's diagBase */
};
static_assert(sizeof(AsmArgs) == ASM_ARGS_BYTES, "argument block");

extern "C" {
/* 0 when the code object is loaded on the device (once per device and process) */
int cpecan_asm_load(int device);
const double *cpecan_asm_coef(int device);
int cpecan_asm_launch_forward(int device, hipStream_t stream, const AsmArgs *args);
int cpecan_asm_launch_backward(int device, hipStream_t stream, const AsmArgs *args);
/* once per batch: context [2] of every alignment (a wave that has done diagonal 0) and the ring's -inf row */
int cpecan_asm_launch_ctx_init(hipStream_t stream, const DevItem *items, long long nItems, char *ctx, long long ctxBytes,
                               double *ring, long long ringDoubles, int ringD);
/* once per batch: the mask table -- per diagonal the lanes of the band per layer (6 dwords), the band's first and last
 * column, and the lanes a ring row is stored / loaded under (6 dwords: the band and the slots next to it on either side) */
int cpecan_asm_launch_masks(hipStream_t stream, const DevItem *items, long long nItems, long long maxDiags, const int *bandTab,
                            unsigned *maskTab);
/* once per run: ring row 0 */
int cpecan_asm_launch_begin(hipStream_t stream, const DevItem *items, long long nItems, double *ring, long long ringDoubles);
}

#endif

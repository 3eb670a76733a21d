/*
 * cpecan_sweep.h -- records shared by the throughput kernels (cpecan_kernel_wave.hip,
 * cpecan_kernel_systolic.hip) and the C-ABI layer that sequences their launches.
 */
#ifndef CPECAN_SWEEP_H_
#define CPECAN_SWEEP_H_

/* Per-alignment state handed between the forward-window and backward-window kernels. */
struct SyState {
    int d;            /* last forward diagonal completed */
    int tracedBackTo; /* as in getPosteriorProbsWithBanding (impl/pairwiseAligner.c:903) */
    int finished;     /* forward reached the last diagonal */
    int bandAi;       /* unused (kept for layout) */
    int winValid, winTop, winFrom, winTo, winAtEnd; /* traceback window for the backward kernel */
    int expectPending; /* Baum-Welch: the window's backward cells are in the B ring, not yet summed */
    long long nPairs, nTot, cells;
};

/* per-window bookkeeping of one totalProbability refresh, kept in HBM scratch (private to the alignment) */
struct WinTotal {
    int t, xmin, xmax, nxmin, nxmax, second;
    double total;
};

#endif

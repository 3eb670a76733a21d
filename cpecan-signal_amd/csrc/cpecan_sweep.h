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
    long long clkShader, clkRef; /* shader-clock and 100 MHz reference ticks the alignment's forward sweeps took (the
                                    ratio is the clock the chip ran at under this load) */
};

/* The wave kernels' record: the forward kernel of launch w describes its window in win[w & 3] while the backward
 * kernel of launch w - 1 is working from the entry before and the post kernel of launch w - 2 may still be reading the
 * one before that (the assembly sweeps' schedule, cpecan_hip.hip: they all run concurrently). */
struct WvWindow {
    int valid; /* 1: described by the forward kernel; 3: swept back, its totals and pairs are the post kernel's to do;
                  2: its candidates could not be trusted, the re-sweep kernel decodes it; 0: done */
    int top, from, to, atEnd;
    int nCand, nRefresh, pad; /* what the sweep back left in scratch for the post kernel */
    double est; /* estimate of the window's totalProbability: the forward cells of its top diagonal dotted with the
                   end vector the sweep back starts from (any fold order; it only steers the candidate test) */
};
struct WvState {
    int d;            /* last forward diagonal completed */
    int tracedBackTo; /* as in getPosteriorProbsWithBanding (impl/pairwiseAligner.c:903) */
    int finished;     /* forward reached the last diagonal */
    int expectPending; /* Baum-Welch: the window's backward cells are in the B ring, not yet summed */
    WvWindow win[4];
    long long nPairs, nTot, cells;
    long long clkShader, clkRef; /* shader-clock and 100 MHz reference ticks the alignment's forward sweeps took (the
                                    ratio is the clock the chip ran at under this load) */
};

/* per-window bookkeeping of one totalProbability refresh, kept in HBM scratch (private to the alignment) */
struct WinTotal {
    int t, xmin, xmax, nxmin, nxmax, second;
    double total;
};

#endif

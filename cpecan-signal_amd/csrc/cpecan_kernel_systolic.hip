/*
 * cpecan_kernel_systolic.hip -- the throughput kernels: banded forward / backward / posterior DP
 * with the anti-diagonal wavefront of the recurrence held in registers.
 *
 * Mapping (designed for CDNA4's 64-lane waves, not translated from anything):
 *   - one 256-thread workgroup per alignment; lanes own reference k-mers, not DP cells: slot
 *     s = x mod 256 is lane s % 64 of wave s / 64 (bands up to 248 k-mers wide).  A k-mer's 17
 *     emission constants stay in its lane's VGPRs while x is inside the band.
 *   - one loop iteration = one anti-diagonal.  Forward: a cell needs (x-1,y) and (x-1,y-1) from the
 *     lane below (one DPP wave-shift per value) and (x,y-1) from itself; only lane 0 of a wave
 *     takes its neighbour from the wave below through 40 bytes of LDS (one LDS-only barrier per
 *     diagonal).  The event a lane scores also moves up one lane per diagonal.
 *   - backward is the mirror image (messages travel down one lane); it is a gather, with the
 *     reference's scatter order of accumulation kept per state (see cpecan_kernel_general.hip).
 *   - forward cells go to HBM once into a per-alignment ring holding one traceback window
 *     ([diagonal][wave][Fm,Fx,Fy,pm,py][lane]) and are read once by the sweep back.
 *   - neither sweep waits for a global load inside its loop: the forward sweep stages its inputs
 *     (events, k-mer rows, band) in LDS every 32 diagonals, the backward sweep fetches ring rows
 *     four diagonals ahead with the loop unrolled by four (see Feed, BandFeed, backward_window).
 *   - the posterior decode works from candidate lists the sweep back collects (backward_window).
 *   - two kernels per traceback window, launched by the C-ABI layer (cpecan_hip.hip): forward to
 *     the next traceback point, then backward + decode of that window; per-alignment state
 *     (SyState) and the ring carry over.
 * MFMA is not used: the recurrence is a scan with an approximate log-add, not a contraction.
 *
 * Numerics: identical to the general kernel and the CPU oracle, bit for bit.  The only algebraic
 * change is the division (x - mu) / sigma, done as a Markstein-corrected multiply by the
 * host-rounded reciprocal (two fused multiply-adds), which returns the correctly rounded quotient
 * (tests/test_systolic_gpu.py checks it against IEEE division on 10^8 operands).
 *
 * -DSY_ABLATE_* and -DSY_PROFILE are timing-study switches (tools/ablate.sh, tools/prof.sh); the
 * ablations compute wrong results by construction and are never built into the product.
 */
#include "cpecan_device.h"
#include "cpecan_sweep.h"

#ifndef SY_R
#define SY_R 4 /* waves per workgroup: 4 (bands up to 248 k-mers) or 3 (up to 184; five workgroups fit a CU) */
#endif
#define SY_P (64 * SY_R)
/* this file is compiled once per SY_R (1..4 waves per workgroup: bands up to 56, 120, 184, 248 k-mers); the symbols
 * of the builds below four carry _r1.._r3, and the pieces that do not depend on SY_R (track, counts, division
 * self-test) exist in the four-wave build only */
#if SY_R == 4
#define SY_SYM(n) n
#elif SY_R == 3
#define SY_SYM(n) n##_r3
#elif SY_R == 2
#define SY_SYM(n) n##_r2
#else
#define SY_SYM(n) n##_r1
#endif
#if SY_R == 1 /* the one-wave backward kernel lands on 129 VGPRs by itself: hold it to four waves per SIMD */
#define SY_BACKWARD_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#ifdef SY_FORCE_WAVES
#define SY_BACKWARD_ATTR __attribute__((amdgpu_waves_per_eu(SY_FORCE_WAVES, SY_FORCE_WAVES)))
#else
#define SY_BACKWARD_ATTR
#endif
#endif
#if SY_R == 4
#define SY_WAVE_OF(x) (((x) >> 6) & 3) /* the wave that owns k-mer x (x >= 0) */
#define SY_WMOD(v) ((v) & 3)           /* a difference of wave indices, any sign, into 0..SY_R-1 */
#define SY_SLOT(x) ((x) & (SY_P - 1))
#else
#define SY_WAVE_OF(x) (((x) >> 6) % SY_R)
#define SY_WMOD(v) ((((v) % SY_R) + SY_R) % SY_R)
#define SY_SLOT(x) ((x) % SY_P)
#endif
#define SY_NPRM 17
#define SY_PREFETCH 4     /* diagonals the backward sweep fetches ahead (== its unroll factor) */
#define SY_CAND_SLACK 0.25 /* candidates: cells within this (log units) below the posterior threshold */
#define SY_CAND_PER_DIAG 4 /* candidate capacity per wave, in records per ring diagonal */
#define SY_DECODE_U 2     /* diagonals per batch of the posterior decode */
#define SY_EXPECT_CHUNKS 8 /* workgroups that share one window's diagonals in the expectation pass */
#define SY_RING_VALUES 5 /* per cell in the forward ring: Fm, Fx, Fy, match emission, gap-Y emission */

#ifdef SY_PROFILE
/* timing build only: cycles per section of the forward step, summed over all waves */
__device__ unsigned long long SY_SYM(sy_prof)[80];
#define sy_prof SY_SYM(sy_prof)
#define PROF_DECL unsigned long long prof_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, tprev_ = __builtin_readcyclecounter(); bool pact_ = false;
#define PROF(k) { const unsigned long long now_ = __builtin_readcyclecounter(); if (pact_) prof_[k] += now_ - tprev_; else prof_[9] += now_ - tprev_; tprev_ = now_; }
#define PROF_ACTIVE(a) { pact_ = (a); if (pact_) prof_[10]++; else prof_[11]++; }
#define PROF_FENCE(x) asm volatile("" : "+v"(x));
#define BPROF_DECL unsigned long long bt_[9]; for (int k_ = 0; k_ < 9; k_++) bt_[k_] = 0; bt_[0] = __builtin_readcyclecounter(); const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#define BPROF(k) bt_[k] = __builtin_readcyclecounter();
#define BPROF_FLUSH if (threadIdx.x == 0) { for (int k_ = 0; k_ < 8; k_++) atomicAdd(&sy_prof[64 + k_], bt_[k_ + 1] - bt_[k_]); atomicMax(&sy_prof[75], bt_[8] - bt_[0]); { const unsigned long long rd_ = __builtin_amdgcn_s_memrealtime() - rt0_; atomicAdd(&sy_prof[76], rd_); atomicMax(&sy_prof[77], rd_); } atomicAdd(&sy_prof[72], 1ull); atomicAdd(&sy_prof[73], (unsigned long long) (sh.scan != 0)); atomicAdd(&sy_prof[74], (unsigned long long) (sh.cnt[1][0][0] + sh.cnt[1][1][0] + sh.cnt[1][2][0] + sh.cnt[1][SY_R - 1][0])); }
#define PROF_FLUSH(wave) if ((threadIdx.x & 63) == 0) { for (int k_ = 0; k_ < 12; k_++) atomicAdd(&sy_prof[(wave) * 16 + k_], prof_[k_]); }
#else
#define PROF_DECL
#define PROF(k)
#define PROF_ACTIVE(a)
#define PROF_FENCE(x)
#define PROF_FLUSH(wave)
#define BPROF_DECL
#define BPROF(k)
#define BPROF_FLUSH
#endif

namespace {

struct Shared {
    double coef[64];         /* lookup() cubics [c3,c2,c1,c0] by n = ceil(2d), 0..15: one ds_read_b128 pair per logAdd */
    double xch[2][SY_R][8];  /* boundary-lane values, double-buffered by diagonal parity */
    double vbuf[SY_P];       /* phase T: per-thread fold results; the sweep's estimate of the total */
    double wbuf[SY_P];       /* decode: partial hit counts */
    double total;
    int cnt[2][SY_R][2];     /* aligned-pair counts per wave, double-buffered */
    int item;
    int scan;                /* this window's posteriors must be decoded by the full scan */
};

/*
 * Inputs of the forward sweep, staged in LDS.  Inside the sweep a wave must never wait on a
 * global load: vmcnt counts loads and stores in one in-order queue, so waiting for any load also
 * waits for the acknowledgement of every ring store issued before it -- a full HBM round trip per
 * anti-diagonal.  Events and k-mer constants are therefore fetched in bulk once every SY_FEED
 * diagonals, for the next two such blocks (both move by at most one index per diagonal), and the
 * sweep itself only reads LDS.
 */
#define SY_FEED 32
#define SY_FEED_EV 128  /* ring of events, by event index            */
#define SY_FEED_ROW 64  /* ring of k-mer constant rows, by k-mer index */
struct Feed {
    double ev[SY_FEED_EV * 2];
    double row[SY_FEED_ROW * CP_ROW];
};

__device__ __forceinline__ double bcast(double v, int srcLane) { /* srcLane wave-uniform */
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srcLane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srcLane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v) {
    const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) v);
    const int hi = __builtin_amdgcn_readfirstlane((int) (v >> 32));
    return ((long long) hi << 32) | lo;
}
/* the work item as wave-uniform (scalar) values: everything derived from it -- band geometry, loop
 * bounds, base addresses -- then stays on the scalar unit instead of occupying vector lanes */
__device__ __forceinline__ DevItem uniform_item(const DevItem &s) {
    DevItem d;
    d.lX = uni64(s.lX); d.lY = uni64(s.lY); d.xOff = uni64(s.xOff); d.yOff = uni64(s.yOff);
    d.anchorOff = uni64(s.anchorOff); d.nAnchors = uni64(s.nAnchors); d.diagBase = uni64(s.diagBase); d.cellBase = 0;
    d.nCells = 0; d.pairBase = uni64(s.pairBase); d.pairCap = uni64(s.pairCap);
    d.totBase = uni64(s.totBase); d.totCap = uni64(s.totCap); d.bwsBase = 0;
    d.model = uni(s.model); d.raggedL = uni(s.raggedL); d.raggedR = uni(s.raggedR); d.maxWidth = 0;
    return d;
}

/* Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain every global
 * store of the forward ring (s_waitcnt vmcnt(0)) on each anti-diagonal; nothing the waves exchange
 * inside an alignment goes through global memory, so only lgkmcnt has to reach zero. */
__device__ __forceinline__ void lds_barrier() {
#ifdef SY_ABLATE_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

/* logAdd (impl/pairwiseAligner.c:238-255), branch-free and bit-identical: hi/lo are the operands
 * as the reference's two branches order them; its "smaller operand is -inf" and ">= 7.5" exits both
 * yield hi, and (-inf) - (-inf) = NaN fails d < 7.5 exactly like those exits.  The cubic's four
 * float-literal coefficients come from a 128-byte LDS table indexed by the piece (<=1, <=2.5,
 * <=4.5, else): two ds_read_b128 instead of a 24-select chain. */
__device__ __forceinline__ double ladd(double x, double y, const double *coef) {
#ifdef SY_ABLATE_LADD
    return x > y ? x : y;
#endif
    double hi, lo;
    /* plain v_max/v_min: operands are never NaN, so the canonicalising pre-ops fmax()/fmin() emit
     * are dead weight in a loop that is bound by instruction issue */
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    const double d = hi - lo;
    /* the piece of lookup() (d <= 1, <= 2.5, <= 4.5, else) from n = ceil(2d): the thresholds are
     * multiples of 1/2 and 2d is exact, so d > 1 <=> n >= 3, d > 2.5 <=> n >= 6, d > 4.5 <=> n >= 10;
     * the LDS table holds the cubic of n's piece at entry n (0..15), so no comparison is needed.
     * NaN (both operands -inf) converts to 0, d >= 7.5 is clamped: either way the cubic is discarded. */
    int n;
#ifdef SY_ABLATE_COEF
    n = 0;
#else
    asm("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(__builtin_ceil(d + d)));
    n = n < 15 ? n : 15;
#endif
    /* one LDS address, two 16-byte reads (the compiler would form two addresses) */
    const unsigned a = (unsigned) (size_t) (const __attribute__((address_space(3))) double *) coef + n * 32u;
    double2 c32, c10;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(c32), "=&v"(c10) : "v"(a));
    const double r = ((c32.x * d + c32.y) * d + c10.x) * d + c10.y + lo;
    return d < 7.5 ? r : hi;
}
__device__ __forceinline__ void init_coef(double *coef) {
    const float t[16] = { -0.009350833524763f, 0.130659527668286f, 0.498799810682272f, 0.693203116424741f,
                          -0.014532321752540f, 0.139942324101744f, 0.495635523139337f, 0.692140569840976f,
                          -0.004605031767994f, 0.063427417320019f, 0.695956496475118f, 0.514272634594009f,
                          -0.000458661602210f, 0.009695946122598f, 0.930734667215156f, 0.168037164329057f };
    /* entry n = ceil(2d) carries the cubic [c3,c2,c1,c0] of the piece d falls in */
    if (threadIdx.x < 64) {
        const int n = threadIdx.x >> 2, piece = n <= 2 ? 0 : n <= 5 ? 1 : n <= 9 ? 2 : 3;
        coef[threadIdx.x] = (double) t[piece * 4 + (threadIdx.x & 3)];
    }
}

/* log N(x; mu, sd) = K + (-0.5*a*a), a = (x-mu)/sd (impl/stateMachine.c:333-343); the quotient is
 * q + fma(-q, sd, t) * rsd with q = t*rsd, rsd = RN(1/sd): Markstein's correction step, which
 * rounds to the same double as the division.  sd == 0 rows carry rsd = 0, K = -inf => -inf. */
__device__ __forceinline__ double lgauss(double x, double mu, double sd, double rsd, double K) {
    const double t = x - mu;
    const double q = t * rsd;
    const double rem = __fma_rn(-q, sd, t);
    const double a = __fma_rn(rem, rsd, q);
    return K + (-0.5 * a * a);
}

/*
 * The band: first and last matrix column (k-mer index) of every anti-diagonal, one int2 per diagonal
 * in HBM, built by the host from band_construct's output (cpecan_hip.hip).  The sweeps read it
 * through a 128-entry LDS ring that the whole workgroup refills every 32 diagonals, so a step costs
 * one LDS read instead of walking the anchor rectangles on the scalar unit.
 */
#define SY_BAND_RING 128
struct BandFeed {
    int2 e[SY_BAND_RING];
};
/* entries of diagonals lo..hi (clipped to 0..D) into the ring; all 256 threads */
__device__ __forceinline__ void band_stage(BandFeed &bf, const int2 *__restrict__ tab, int D, int lo, int hi) {
    for (int d = lo + (int) threadIdx.x; d <= hi; d += SY_P)
        if (d >= 0 && d <= D) bf.e[d & (SY_BAND_RING - 1)] = tab[d];
}
__device__ __forceinline__ void band_get(const BandFeed &bf, int d, int &xmin, int &xmax) {
    const int2 v = bf.e[d & (SY_BAND_RING - 1)];
    xmin = uni(v.x);
    xmax = uni(v.y);
}
/* straight from HBM (start-up paths only) */
__device__ __forceinline__ void band_load(const int2 *__restrict__ tab, int d, int &xmin, int &xmax) {
    const int2 v = tab[d > 0 ? d : 0];
    xmin = uni(v.x);
    xmax = uni(v.y);
}

/* does wave w hold an in-band slot: waves (xmin>>6) .. (xmax>>6), modulo R */
__device__ __forceinline__ bool row_active(int w, int xmin, int xmax) {
    int first = xmin >> 6, n = (xmax >> 6) - first;
    return SY_WMOD(w - first) <= n;
}

/* logAdd-fold of one value per lane into acc (wave-uniform in and out), lanes in ascending order;
 * visits only the lanes that can change the running value (cp_wave_seq_fold with the LDS-table
 * logAdd, so that no coefficient constants occupy registers around the call) */
__device__ __forceinline__ double wave_fold(double acc, double v, const double *coef) {
    const int lane = threadIdx.x & 63;
    unsigned long long after = ~0ull;
#pragma unroll 1
    for (;;) {
        const bool eff = ((after >> lane) & 1ull) && (v > CP_NEG_INF) && !(acc - v >= 7.5);
        const unsigned long long m = __ballot(eff);
        if (m == 0ull) break;
        const int first = __ffsll((long long) m) - 1;
        acc = ladd(acc, bcast(v, first), coef);
        after = first >= 63 ? 0ull : (~0ull << (first + 1));
    }
    return acc;
}

struct ItemOut {
    long long *pairs;
    double *logp;
    long long pairCap;
    long long *totXay;
    double *totVal;
    long long totCap;
    long long nPairs, nTot;
};

__device__ __forceinline__ void load_params(double (&dst)[SY_NPRM], const double *__restrict__ track, int x) {
    const double *p = track + (long long) x * CP_ROW;
#pragma unroll
    for (int j = 0; j < SY_NPRM; j++) dst[j] = p[j];
}
__device__ __forceinline__ double set_lane(double v, int dst, double x) { /* x, dst wave-uniform */
    if ((int) (threadIdx.x & 63) == dst) v = x;
    return v;
}

/* lane i <- lane i-1 of src, lane 0 <- old (DPP wave_shr:1 leaves lanes without a source untouched) */
__device__ __forceinline__ double shr1(double old, double src) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x138, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
/* lane i <- lane i+1 of src, lane 63 <- old */
__device__ __forceinline__ double shl1(double old, double src) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), 0x130, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

/* a load that cannot be served from a line this CU cached before another wave (or an earlier phase
 * of the same workgroup) rewrote it */
template <typename V> __device__ __forceinline__ V ld_agent(V *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Geometry {
    int lane, wave, waveBelow, waveAbove;
    double *rw, *rwb; /* ring bases of this lane's slot and of the slot below it */
    int ringMask;
    __device__ __forceinline__ double *rp(int d, int s) const {
        return rw + (long long) (d & ringMask) * (SY_R * SY_RING_VALUES * 64) + s * 64;
    }
    __device__ __forceinline__ double *rpb(int d, int s) const {
        return rwb + (long long) (d & ringMask) * (SY_R * SY_RING_VALUES * 64) + s * 64;
    }
};

__device__ __forceinline__ Geometry make_geometry(double *ring, int ringD) {
    Geometry g;
    g.lane = threadIdx.x & 63;
    g.wave = uni(threadIdx.x >> 6);
    g.waveBelow = (g.wave + SY_R - 1) % SY_R;
    g.waveAbove = (g.wave + 1) % SY_R;
    /* ring of forward diagonals: [diagonal & (ringD-1)][wave][Fm,Fx,Fy,pm,py][lane] */
    g.rw = ring + g.wave * (SY_RING_VALUES * 64) + g.lane;
    g.rwb = ring + (g.lane == 0 ? g.waveBelow : g.wave) * (SY_RING_VALUES * 64) + ((g.lane + 63) & 63);
    g.ringMask = ringD - 1;
    return g;
}

/* The next traceback point (:917-921) from the band alone: the first diagonal above dAfter that is at
 * least dMin and narrow enough, or the last diagonal D.  Whole workgroup, 256 diagonals per round. */
__device__ int next_traceback_point(const int2 *__restrict__ tab, int D, int dAfter, long long dMin,
                                    long long widthLimit, Shared &sh) {
    long long b0 = dAfter + 1 > dMin ? dAfter + 1 : dMin;
    if (b0 >= D) return D;
    int base = (int) b0;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) sh.item = 0x7fffffff;
        __syncthreads();
        const int d = base + (int) threadIdx.x;
        if (d >= D) atomicMin(&sh.item, D);
        else {
            const int2 v = tab[d];
            if ((long long) (v.y - v.x + 1) <= widthLimit) atomicMin(&sh.item, d);
        }
        __syncthreads();
        const int r = uni(sh.item);
        if (r != 0x7fffffff) return r;
        base += SY_P;
    }
}

/* Forward sweep of one alignment from its saved diagonal up to (and including) the next traceback
 * point; describes the window for the backward kernel. */
__device__ void forward_window(const DevItem &it, const DevParams &P, const int2 *__restrict__ bandTab,
                               const double *__restrict__ track, const double *__restrict__ events,
                               const double *__restrict__ model, double *ring, int ringD,
                               SyState *state, Shared &sh, Feed &fd, BandFeed &bf) {
    const Geometry g = make_geometry(ring, ringD);
    const int lane = g.lane, wave = g.wave;
    const int lX = (int) it.lX, lY = (int) it.lY, D = lX + lY;
    const double *__restrict__ ev = events + 3 * it.yOff;
    const double *cf = sh.coef;
    /* a -inf gapY->gapX transition (the nanopore default, stateMachine.c:1287) contributes
     * logAdd(acc, -inf) == acc: skip that term (wave-uniform) */
    const bool hasSwitchX = model[T_GAP_SWITCH_TO_X] > CP_NEG_INF;
    double T[9];
#pragma unroll
    for (int i = 0; i < 9; i++) T[i] = model[i];

    const int d0 = uni(ld_agent(&state->d));
    int tracedBackTo = uni(ld_agent(&state->tracedBackTo));
    long long cells = uni64(ld_agent(&state->cells));

    /* ---- per-slot state (this lane's k-mer) ---- */
    int xs;
    double prm[SY_NPRM];
    double Fm, Fx, Fy; /* forward cell, current diagonal   */
    double Lm, Lx, Ly; /* slot-1's cell, previous diagonal */
    double em, en;     /* event scored on this diagonal    */
    int xin, xminP;

    if (d0 == 0) {
        /* diagonal 0: the single cell (0,0) holds the start vector (:897-898, stateMachine.c:1168-1177) */
        xs = wave * 64 + lane;
#pragma unroll
        for (int j = 0; j < SY_NPRM; j++) prm[j] = 0.0;
        Fm = Fx = Fy = Lm = Lx = Ly = CP_NEG_INF;
        em = en = 0.0;
        if (wave == 0 && lane == 0) {
            load_params(prm, track, 0);
            Fm = it.raggedL ? CP_NEG_INF : 0.0;
            Fx = it.raggedL ? 0.0 : CP_NEG_INF;
            Fy = Fx;
        }
        if (wave == 0 && lane == 0) { /* only cells of the band are ever stored to the ring */
            *g.rp(0, 0) = Fm;
            *g.rp(0, 1) = Fx;
            *g.rp(0, 2) = Fy;
            *g.rp(0, 3) = 0.0;
            *g.rp(0, 4) = 0.0;
        }
        cells += 1;
        xin = 1;
        xminP = 0;
    } else {
        /* resume at d0: constants of the k-mers in the band, forward cells of d0 and d0-1, events */
        int xmin, xmax, qmin, qmax;
        band_load(bandTab, d0 - 1, qmin, qmax);
        band_load(bandTab, d0, xmin, xmax);
        xs = wave * 64 + lane;
        xs += ((xmin - xs + SY_P - 1) / SY_P) * SY_P; /* the k-mer >= xmin that lives in this slot */
        const bool v = xs <= xmax;
        load_params(prm, track, xs <= lX ? xs : lX);
        /* ring slots of cells outside the band hold stale data: mask per lane */
        Fm = v ? *g.rp(d0, 0) : CP_NEG_INF;
        Fx = v ? *g.rp(d0, 1) : CP_NEG_INF;
        Fy = v ? *g.rp(d0, 2) : CP_NEG_INF;
        const bool a1 = xs - 1 >= qmin && xs - 1 <= qmax;
        Lm = a1 ? *g.rpb(d0 - 1, 0) : CP_NEG_INF;
        Lx = a1 ? *g.rpb(d0 - 1, 1) : CP_NEG_INF;
        Ly = a1 ? *g.rpb(d0 - 1, 2) : CP_NEG_INF;
        const int ei = d0 - xs - 1;
        const bool okE = v && ei >= 0 && ei < lY;
        em = okE ? ev[3 * (long long) ei] : 0.0;
        en = okE ? ev[3 * (long long) ei + 1] : 0.0;
        xin = xmax + 1;
        xminP = xmin;
    }
    int evHi = d0 - xminP - 1, rowHi = xin; /* first event / k-mer row not yet staged */

    /*
     * Which diagonals need all three states in the ring.  The sweep back reads only the match cell of
     * a diagonal, except where it refreshes totalProbability (every 10th decoded diagonal, counted
     * down from the first one of ITS window: it then reads every state of that diagonal and of the
     * one below), and this sweep resumes from the last two diagonals of a launch.  Where the windows
     * will start is a function of the band alone, so it is known here: this launch ends at topW; its
     * diagonals up to fromW are decoded by window W (first decoded diagonal tpA), the ones above by
     * the next window (tpB).  Everywhere else the two gap states are not stored: 16 of the 40 bytes
     * a cell costs on the way up.
     */
    const long long widthLimit = P.expansion * 2 + 1;
    const int topW = next_traceback_point(bandTab, D, d0, tracedBackTo + P.minDiags, widthLimit, sh);
    const bool endW = topW == D;
    const int fromW = topW - (endW ? 0 : (int) P.tbDiags + 1);
    const int tpA = topW < fromW ? topW : fromW;
    int tpB = tpA;
    bool allFull = false;
    if (!endW) {
        const int topN = next_traceback_point(bandTab, D, topW, fromW + P.minDiags, widthLimit, sh);
        const int fromN = topN - (topN == D ? 0 : (int) P.tbDiags + 1);
        tpB = topN < fromN ? topN : fromN;
        allFull = tpB < topW; /* windows shorter than the traceback margin: keep everything */
    }
    if (P.mode != 0) allFull = true; /* the expectation pass reads every state of every diagonal */
    /* (tpA - d) mod 10 and (tpB - d) mod 10 for the diagonal being computed, kept incrementally */
    /* (tpA - d) mod 10 and (tpB - d) mod 10 for the next diagonal whose mask bit is computed */
    int rA = ((tpA - (d0 + 1)) % 10 + 10) % 10, rB = endW ? 0x40000000 : ((tpB - (d0 + 1)) % 10 + 10) % 10;
    const int fullFrom = allFull ? -0x40000000 : topW - 1;
    if (lane == 63) {
        double *x = sh.xch[d0 & 1][wave];
        x[0] = Fm; x[1] = Fx; x[2] = Fy; x[3] = em; x[4] = en;
    }

    const long long tbFromL = tracedBackTo + P.minDiags;
    const int tbFrom = uni((int) (tbFromL < 0x7fffffff ? tbFromL : 0x7fffffff));
    const int tbWidth = uni((int) (widthLimit < 0x7fffffff ? widthLimit : 0x7fffffff));
    PROF_DECL
    /* blocks of SY_FEED diagonals: the loads sit between the blocks, the inner loop has none */
#pragma unroll 1
    for (int db = d0 + 1; db <= D; db += SY_FEED) {
        {
            /* stage what the next two blocks of diagonals can ask for: the top cell's event index
             * d-xmin-1 and the entering k-mer xin each advance by at most one per diagonal */
            int fxmin, fxmax;
            band_load(bandTab, db, fxmin, fxmax);
            band_stage(bf, bandTab, D, db, db + 2 * SY_FEED - 1);
            const int evTo = db - fxmin - 1 + 3 * SY_FEED, rowTo = xin + 2 * SY_FEED;
#pragma unroll 1
            for (int i = evHi * 2 + (int) threadIdx.x; i < evTo * 2; i += SY_P) {
                const int e = i >> 1;
                fd.ev[(i & (2 * SY_FEED_EV - 1))] = e >= 0 && e < lY ? ev[3 * (long long) e + (i & 1)] : 0.0;
            }
#pragma unroll 1
            for (int i = rowHi * CP_ROW + (int) threadIdx.x; i < rowTo * CP_ROW; i += SY_P) {
                const int x = i / CP_ROW, j = i - x * CP_ROW;
                fd.row[(x & (SY_FEED_ROW - 1)) * CP_ROW + j] = track[(long long) (x <= lX ? x : lX) * CP_ROW + j];
            }
            evHi = evTo;
            rowHi = rowTo;
            __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0): nothing pending past this point */
        }
        const int dbEnd = db + SY_FEED - 1 < D ? db + SY_FEED - 1 : D;
        unsigned fullMask = 0u; /* bit j: diagonal db + j keeps all three states */
#pragma unroll 1
        for (int j = 0; j < SY_FEED; j++) {
            const int dj = db + j;
            const int rHere = dj <= fromW ? rA : rB, rAbove = dj + 1 <= fromW ? rA : rB;
            fullMask |= (dj >= fullFrom || rHere == 0 || rAbove == 1 ? 1u : 0u) << j;
            rA = rA == 0 ? 9 : rA - 1;
            rB = rB == 0 ? 9 : rB - 1;
        }
        fullMask = (unsigned) uni((int) fullMask);
#pragma unroll 1
        for (int d = db; d <= dbEnd; d++) {
        PROF(0)
        PROF(1)
        lds_barrier(); /* also orders the band/event/k-mer staging above before the reads below */
        int xmin, xmax;
#ifdef SY_ABLATE_BAND
        xmin = d / 3; xmax = xmin + 132;
#else
        band_get(bf, d, xmin, xmax);
#endif
        cells += xmax - xmin + 1;
        const bool full = ((fullMask >> (d - db)) & 1u) != 0u;
        PROF_ACTIVE(row_active(wave, xmin, xmax))
        PROF(2)
#ifdef SY_ABLATE_XCH
        const double rm = shr1(Fm, Fm), rx = shr1(Fx, Fx), ry = shr1(Fy, Fy);
        em = shr1(em, em);
        en = shr1(en, en);
#else
        const double *xb = sh.xch[(d - 1) & 1][g.waveBelow];
        const double rm = shr1(xb[0], Fm), rx = shr1(xb[1], Fx), ry = shr1(xb[2], Fy);
        em = shr1(xb[3], em);
        en = shr1(xb[4], en);
#endif
        PROF_FENCE(em) PROF_FENCE(en)
        PROF(3)
        if (xs < xmin) xs += SY_P;
        const bool valid = xs <= xmax;
        while (xin <= xmax) { /* the entering k-mer's constants (at most one k-mer per step) */
#ifdef SY_ABLATE_INSTALL
            if (d == -1) {
#else
            if (SY_WAVE_OF(xin) == wave && lane == (xin & 63)) {
#endif
                const double *r = fd.row + (xin & (SY_FEED_ROW - 1)) * CP_ROW;
#pragma unroll
                for (int j = 0; j < SY_NPRM; j++) prm[j] = r[j];
            }
            xin++;
        }
        if (xmin == xminP && SY_WAVE_OF(xmin) == wave && lane == (xmin & 63)) {
            /* the top cell's event is new.  Index -1 is NULLEVENT (:261): its emissions only ever
             * meet -inf cells, the staged 0.0 keeps NaN out */
            const double *e = fd.ev + ((2 * (d - xmin - 1)) & (2 * SY_FEED_EV - 1));
            em = e[0];
            en = e[1];
        }
        PROF_FENCE(em) PROF_FENCE(en) PROF_FENCE(prm[0])
        PROF(4)
        double nmv = CP_NEG_INF, nxv = CP_NEG_INF, nyv = CP_NEG_INF;
        if (row_active(wave, xmin, xmax)) {
            const double px = prm[CP_GAPX];
#ifdef SY_ABLATE_EMIT
            double pm = em + prm[CP_K1], py = en + prm[CP_YK1];
#else
            double pm = lgauss(em, prm[CP_MU], prm[CP_SD], prm[CP_RSD], prm[CP_K1])
                            + lgauss(en, prm[CP_NMU], prm[CP_NSD], prm[CP_RNSD], prm[CP_K2]);
            double py = lgauss(em, prm[CP_YMU], prm[CP_YSD], prm[CP_RYSD], prm[CP_YK1])
                            + lgauss(en, prm[CP_YNMU], prm[CP_YNSD], prm[CP_RYNSD], prm[CP_YK2]);
#endif
            PROF_FENCE(pm) PROF_FENCE(py)
            PROF(5)
            /* cell_calculateForward: to[t] = logAdd(to[t], from[f] + (eP + tP)) (:365-376) in the
             * order of stateMachine3_cellCalculate (stateMachine.c:1314-1333) */
            double gx = rm + (px + T[T_GAP_OPEN_X]);
            gx = ladd(gx, rx + (px + T[T_GAP_EXTEND_X]), cf);
            if (hasSwitchX) gx = ladd(gx, ry + (px + T[T_GAP_SWITCH_TO_X]), cf);
            double mm = Lm + (pm + T[T_MATCH_CONTINUE]);
            mm = ladd(mm, Lx + (pm + T[T_MATCH_FROM_GAP_X]), cf);
            mm = ladd(mm, Ly + (pm + T[T_MATCH_FROM_GAP_Y]), cf);
            double gy = Fm + (py + T[T_GAP_OPEN_Y]);
            gy = ladd(gy, Fy + (py + T[T_GAP_EXTEND_Y]), cf);
            PROF_FENCE(mm) PROF_FENCE(gx) PROF_FENCE(gy)
            PROF(6)
            nmv = valid ? mm : CP_NEG_INF;
            nxv = valid ? gx : CP_NEG_INF;
            nyv = valid ? gy : CP_NEG_INF;
#ifdef SY_ABLATE_STORE
            if (valid && d == -1) {
#else
            if (valid) { /* cells outside the band cost no HBM traffic */
#endif
                *g.rp(d, 0) = mm;
                if (full) {
                    *g.rp(d, 1) = gx;
                    *g.rp(d, 2) = gy;
                }
                *g.rp(d, 3) = pm; /* the sweep back re-uses the two event-dependent emissions */
                *g.rp(d, 4) = py;
            }
        }
        PROF(7)
        Lm = rm; Lx = rx; Ly = ry;
        Fm = nmv; Fx = nxv; Fy = nyv;
        if (lane == 63) {
            double *x = sh.xch[d & 1][wave];
            x[0] = Fm; x[1] = Fx; x[2] = Fy; x[3] = em; x[4] = en;
        }
        xminP = xmin;
        PROF(8)

        const bool atEnd = d == D;
        const bool tb = d >= tbFrom && xmax - xmin < tbWidth; /* both wave-uniform 32-bit: scalar compares */
        if (atEnd || tb) { /* traceback point (:917-921): hand the window to the backward kernel */
            if (threadIdx.x == 0) {
                const int from = d - (atEnd ? 0 : (int) P.tbDiags + 1);
                state->d = d;
                state->finished = atEnd ? 1 : 0;
                state->winValid = 1;
                state->winTop = d;
                state->winFrom = from;
                state->winTo = tracedBackTo;
                state->winAtEnd = atEnd ? 1 : 0;
                state->tracedBackTo = from;
                state->cells = cells;
            }
            PROF_FLUSH(wave)
            return;
        }
        }
    }
}

/*
 * Backward sweep + posterior decode of one traceback window (:921-992), in three phases:
 *  S  the sweep back: one anti-diagonal per iteration, cells and messages in registers.  It leaves
 *     per cell F.match + B.match (in the ring slot of the match emission, dead by then) and, on the
 *     diagonals where the reference refreshes totalProbability, the two per-cell terms of that sum
 *     (HBM scratch);
 *  T  totalProbability (:736-754) for every refresh of the window at once: the reference's
 *     order-dependent logAdd fold is inherently serial, so each THREAD folds one diagonal's terms
 *     privately, in the reference's order -- ~200 independent chains instead of one wave-wide chain
 *     per refresh inside the sweep;
 *  D  diagonalCalculationPosteriorMatchProbs (:756-795) for the window: hits are counted per
 *     diagonal, prefix-summed in emission order (diagonals descending, x-y ascending) and written.
 */
__device__ void backward_window(const DevItem &it, const DevParams &P, const int2 *__restrict__ bandTab,
                                const double *__restrict__ track, const double *__restrict__ model,
                                double *ring, int ringD, SyState *state, ItemOut &out, Shared &sh,
                                BandFeed &bf, int *cntBuf, WinTotal *wtot, double *vw,
                                unsigned long long *msk, int2 *candKx, double *candFb, double *bring) {
    const Geometry g = make_geometry(ring, ringD);
    const int lane = g.lane, wave = g.wave;
    const int lX = (int) it.lX, D = (int) (it.lX + it.lY);
    const double *cf = sh.coef;
    const bool hasSwitchX = model[T_GAP_SWITCH_TO_X] > CP_NEG_INF;
    double T[9];
#pragma unroll
    for (int i = 0; i < 9; i++) T[i] = model[i];

    const int dTop = uni(ld_agent(&state->winTop)), tracedBackFrom = uni(ld_agent(&state->winFrom)),
              tracedBackTo = uni(ld_agent(&state->winTo));
    const bool atEnd = uni(ld_agent(&state->winAtEnd)) != 0;
    BPROF_DECL
    const int tPost0 = dTop < tracedBackFrom ? dTop : tracedBackFrom; /* first decoded diagonal */
    const int nPost = tPost0 - tracedBackTo;                         /* diagonals decoded      */
    band_stage(bf, bandTab, D, dTop - (SY_BAND_RING - 1), dTop);
    if (threadIdx.x == 0) sh.scan = 0;
    __syncthreads();
    /* Candidates for the posterior decode, collected by the sweep: a cell whose F.match + B.match
     * lies within SY_CAND_SLACK of the threshold, measured against an estimate of the window's
     * totalProbability taken at its first refresh.  Phase T checks every exact total of the window
     * against the estimate; if one strays (or a list overflows) the window is decoded by the scan. */
    const int candCap = SY_CAND_PER_DIAG * ringD;
    int2 *const myKx = candKx + (long long) wave * candCap;
    double *const myFb = candFb + (long long) wave * candCap;
    int nCand = 0;
    double candThr = CP_NEG_INF, totEst = CP_NEG_INF;

    /* ------------------------------ phase S: the sweep back ------------------------------ */
    {
        int bxmin, bxmax;                 /* band of diagonal t   */
        band_get(bf, dTop, bxmin, bxmax);
        int nxmin = bxmin, nxmax = bxmax; /* band of diagonal t+1 */
        int pxmin, pxmax;                 /* band of diagonal t-1 */
        band_get(bf, dTop - 1, pxmin, pxmax);

        int xs = wave * 64 + lane;        /* backward representative: xmax-P < x <= xmax */
        xs += ((bxmin - xs + SY_P - 1) / SY_P) * SY_P;
        if (xs > bxmax) xs -= SY_P;
        bool tvalid = xs >= bxmin;        /* slot in band on diagonal t */
        double Bm, Bx, By;                                        /* backward cell on diagonal t        */
        /* What a cell hands to the diagonals below are sums B + (eP + tP) of its own backward value, its own
         * emission and a transition.  The slot above sends the two ingredients (B.match with its match emission,
         * B.gapX with its gap-X emission) down the lanes and the receiver forms the sums: four values to shift and
         * to hold instead of eight; the sums are the same expressions, so the doubles are the same. */
        double hB = CP_NEG_INF, hP = 0.0;        /* B.match and match emission of slot+1 on t+2 (middle block) */
        double Um = CP_NEG_INF, Uy = CP_NEG_INF; /* upper-block sums from t+1 (same slot)                       */
        double pmPrev = 0.0, BmPrev = CP_NEG_INF;                 /* match emission / backward match of t+1 */
        {
            double e0, e1, e2; /* end state vector (stateMachine.c:1179-1207) */
            if (atEnd && it.raggedR) {
                e0 = (T[T_GAP_OPEN_X] + T[T_GAP_OPEN_Y]) / 2.0;
                e1 = T[T_GAP_EXTEND_X];
                e2 = T[T_GAP_EXTEND_Y];
            } else {
                e0 = T[T_MATCH_CONTINUE];
                e1 = T[T_MATCH_FROM_GAP_X];
                e2 = T[T_MATCH_FROM_GAP_Y];
            }
            Bm = tvalid ? e0 : CP_NEG_INF;
            Bx = tvalid ? e1 : CP_NEG_INF;
            By = tvalid ? e2 : CP_NEG_INF;
        }
        /* the sweep needs one constant per k-mer, its gap-X emission: held per slot, refreshed from
         * a 64-k-mer chunk when a k-mer enters at the low edge of the band */
        double pxReg = track[(long long) (tvalid ? xs : 0) * CP_ROW + CP_GAPX];
        int xinB = bxmin - 1; /* k-mers above xinB are installed */
        int pxBase = (xinB >= 0 ? xinB : 0) & ~63;
        double pxChunk = track[(long long) min(pxBase + lane, lX) * CP_ROW + CP_GAPX];

        /*
         * The forward match cell and the two emissions of a diagonal are fetched SY_PREFETCH diagonals
         * before the sweep reaches it: a wave that waits for a load also waits for everything issued
         * before it (vmcnt is one in-order queue), so a fetch consumed one diagonal later costs a full
         * HBM round trip per diagonal.  The loop is unrolled by the prefetch depth so that each
         * in-flight diagonal has registers of its own and the waits are exact counts.  The loads are
         * unconditional (no branch, so the counts hold): a lane whose slot cannot be in the band on
         * that diagonal reads a fixed dummy line instead, and every value is masked when consumed
         * (only cells of the band are ever stored to the ring; other slots hold stale data).
         */
        auto fetch = [&](const int tt, const bool want, double &f, double &pm, double &py)
                         __attribute__((always_inline)) {
            const double *p = want ? g.rp(tt, 0) : g.rw;
            f = p[0];
            pm = p[3 * 64];
            py = p[4 * 64];
        };
        int calcs = 0, nTotWin = 0;
        int xsN = xs; /* this slot's k-mer on diagonal t+1 */
        auto step = [&](const int t, double &qF, double &qPm, double &qPy) __attribute__((always_inline)) {
            const bool active = row_active(wave, bxmin, bxmax);
            const bool activeN = row_active(wave, nxmin, nxmax);
            if (t < dTop) {
                xsN = xs;
                if (xs > bxmax) xs -= SY_P;
            }
            const bool vt = xs >= bxmin;
            const double fMc = vt ? qF : CP_NEG_INF, pmc = vt ? qPm : 0.0, pyc = vt ? qPy : 0.0;
            /* the slot's k-mer SY_PREFETCH diagonals down is xs or out of band (bands <= 256 - depth) */
            fetch(t - SY_PREFETCH, xs >= bxmin - SY_PREFETCH, qF, qPm, qPy);
            if (t < dTop) {
                lds_barrier();
                const double *xa = sh.xch[(t + 1) & 1][g.waveAbove];
                /* of slot+1 on t+1: B.gapX with its k-mer's gap-X emission (lower block of t+1), B.match with its
                 * match emission (middle block, used one diagonal further down) */
                const double rBx = shl1(xa[0], Bx), rPx = shl1(xa[1], pxReg);
                const double rBm = shl1(xa[2], Bm), rPm = shl1(xa[3], pmPrev);
                const bool bvalid = xs >= bxmin;
                while (xinB >= bxmin) {
                    if (xinB < pxBase) {
                        pxBase -= 64;
                        pxChunk = track[(long long) min(pxBase + lane, lX) * CP_ROW + CP_GAPX];
                    }
                    const double v = bcast(pxChunk, xinB - pxBase);
                    if (SY_WAVE_OF(xinB) == wave) pxReg = set_lane(pxReg, xinB & 63, v);
                    xinB--;
                }
                BmPrev = Bm;
                tvalid = bvalid;
                double bm = CP_NEG_INF, bx = CP_NEG_INF, by = CP_NEG_INF;
                if (active) {
                    /* gather form of cell_calculateBackward: (t+2) middle block, then (t+1, smaller
                     * x-y) upper block, then (t+1, larger x-y) lower block */
                    bm = ladd(ladd(hB + (hP + T[T_MATCH_CONTINUE]), Um, cf), rBx + (rPx + T[T_GAP_OPEN_X]), cf);
                    bx = ladd(hB + (hP + T[T_MATCH_FROM_GAP_X]), rBx + (rPx + T[T_GAP_EXTEND_X]), cf);
                    by = ladd(hB + (hP + T[T_MATCH_FROM_GAP_Y]), Uy, cf);
                    if (hasSwitchX) by = ladd(by, rBx + (rPx + T[T_GAP_SWITCH_TO_X]), cf);
                    bm = bvalid ? bm : CP_NEG_INF;
                    bx = bvalid ? bx : CP_NEG_INF;
                    by = bvalid ? by : CP_NEG_INF;
                }
                Bm = bm; Bx = bx; By = by;
                hB = rBm; hP = rPm;
            }
            /* what this diagonal hands down: the upper-block sums stay in the slot (a cell outside the band has
             * B = -inf and a zero emission: the sums are -inf by themselves); the rest leaves as B and emission */
            Um = By + (pyc + T[T_GAP_OPEN_Y]);
            Uy = By + (pyc + T[T_GAP_EXTEND_Y]);
            if (lane == 0) {
                double *x = sh.xch[t & 1][wave];
                x[0] = Bx; x[1] = pxReg; x[2] = Bm; x[3] = pmc;
            }

            if (t <= tracedBackFrom) {
                const double fb = fMc + Bm;
                if (calcs++ % 10 == 0) {
                    /* per-cell terms of diagonalCalculationTotalProbability (:736-754), folded in
                     * phase T: v = cell_dotProduct(forward[t], backward[t]) (:391-397) ... */
                    const bool second = t + 1 <= dTop;
                    double v = CP_NEG_INF, w = CP_NEG_INF;
                    /* all five loads of a refresh go out together (one HBM round trip, not two: the store of
                     * v below would otherwise fence the second group behind the first); masked when used */
                    const bool below = second && activeN && xsN - 1 >= pxmin && xsN - 1 <= pxmax;
#ifdef SY_ABLATE_TOTLOADS
                    const double fx = fMc, fy = fMc, r0 = fMc, r1 = fMc, r2 = fMc;
#else
                    const double *pa = tvalid ? g.rp(t, 1) : g.rw, *pb = below ? g.rpb(t - 1, 0) : g.rw;
                    const double fx = pa[0], fy = pa[64], r0 = pb[0], r1 = pb[64], r2 = pb[128];
#endif
                    if (tvalid) {
                        v = fb;
                        v = ladd(v, fx + Bx, cf);
                        v = ladd(v, fy + By, cf);
                        vw[((long long) nTotWin * 2 + 0) * SY_P + wave * 64 + lane] = v;
                    }
                    if (second && activeN) {
                        /* ... and w = matches stepping over t: forward[t-1] --match--> the cells of
                         * t+1, dotted with backward[t+1] (only the match state of that clone is
                         * ever above -inf) */
                        const double s0 = below ? r0 : CP_NEG_INF, s1 = below ? r1 : CP_NEG_INF,
                                     s2 = below ? r2 : CP_NEG_INF;
                        double mm = s0 + (pmPrev + T[T_MATCH_CONTINUE]);
                        mm = ladd(mm, s1 + (pmPrev + T[T_MATCH_FROM_GAP_X]), cf);
                        mm = ladd(mm, s2 + (pmPrev + T[T_MATCH_FROM_GAP_Y]), cf);
                        w = mm + BmPrev;
                        vw[((long long) nTotWin * 2 + 1) * SY_P + wave * 64 + lane] = w;
                    }
                    if (nTotWin == 0) {
                        /* the estimate: the same terms folded in any order (it only steers the
                         * candidate test; the exact, ordered folds are phase T's) */
                        double acc = wave_fold(CP_NEG_INF, v, cf);
                        acc = wave_fold(acc, w, cf);
                        if (lane == 0) sh.vbuf[wave] = acc;
                        lds_barrier();
                        double est = sh.vbuf[0];
#pragma unroll
                        for (int q = 1; q < SY_R; q++) est = ladd(est, sh.vbuf[q], cf);
                        totEst = est;
                        candThr = est + (P.logThrSlack - SY_CAND_SLACK);
                    }
                    if (threadIdx.x == 0) {
                        WinTotal w;
                        w.t = t; w.xmin = bxmin; w.xmax = bxmax; w.nxmin = nxmin; w.nxmax = nxmax;
                        w.second = second ? 1 : 0;
                        w.total = CP_NEG_INF;
                        wtot[nTotWin] = w;
                    }
                    nTotWin++;
                }
                /* exponent of the posterior, less the total: parked in the emission slot this diagonal
                 * no longer needs (the forward cells themselves stay intact: the next window's
                 * refresh at its lowest diagonal reads forward[tracedBackFrom], :944,:985) */
                if (P.mode != 0) {
                    /* Baum-Welch: the backward cell goes to its own ring for the expectation kernel
                     * (the emissions in slots 3, 4 stay: that kernel re-uses them) */
                    if (tvalid) {
                        double *b = bring + (long long) (t & g.ringMask) * (SY_R * 3 * 64) + wave * (3 * 64) + lane;
                        b[0] = Bm;
                        b[64] = Bx;
                        b[128] = By;
                    }
                }
#ifndef SY_ABLATE_FB
                else if (tvalid) *g.rp(t, 3) = fb;
#endif
                const bool cand = P.mode == 0 && tvalid && fb >= candThr && fb > CP_NEG_INF;
                const unsigned long long cm = __ballot(cand);
                if (cm != 0ull) {
                    const int ci = nCand + __popcll(cm & ((1ull << lane) - 1ull));
                    if (cand && ci < candCap) {
                        myKx[ci] = make_int2(tPost0 - t, xs);
                        myFb[ci] = fb;
                    }
                    nCand += __popcll(cm);
                }
            }
            pmPrev = pmc;
            nxmin = bxmin; nxmax = bxmax;
            bxmin = pxmin; bxmax = pxmax;
            if (t - 2 >= tracedBackTo) band_get(bf, t - 2, pxmin, pxmax);
        };
        double q0F, q0Pm, q0Py, q1F, q1Pm, q1Py, q2F, q2Pm, q2Py, q3F, q3Pm, q3Py;
        {
            const bool want = xs >= bxmin - SY_PREFETCH;
            fetch(dTop, want, q0F, q0Pm, q0Py);
            fetch(dTop - 1, want, q1F, q1Pm, q1Py);
            fetch(dTop - 2, want, q2F, q2Pm, q2Py);
            fetch(dTop - 3, want, q3F, q3Pm, q3Py);
        }
        int t = dTop, bandLo = dTop - (SY_BAND_RING - 1); /* lowest diagonal whose band is staged */
#pragma unroll 1
        for (; t - (SY_PREFETCH - 1) > tracedBackTo; t -= SY_PREFETCH) {
            if (t - 64 < bandLo) { /* 32 more band entries, long before the sweep reads them */
                band_stage(bf, bandTab, D, bandLo - 32, bandLo - 1);
                bandLo -= 32;
            }
            step(t, q0F, q0Pm, q0Py);
            step(t - 1, q1F, q1Pm, q1Py);
            step(t - 2, q2F, q2Pm, q2Py);
            step(t - 3, q3F, q3Pm, q3Py);
        }
        /* the last diagonals of the window (fewer than the prefetch depth): fetched on the spot */
#pragma unroll 1
        for (; t > tracedBackTo; t--) {
            double f, pm, py;
            fetch(t, true, f, pm, py);
            step(t, f, pm, py);
        }
        sh.item = nTotWin;
        if (lane == 0) sh.cnt[1][wave][0] = nCand;
        if (nCand > candCap) sh.scan = 1;
    }
    __syncthreads(); /* the ring rows written above are read by other waves below */
    const int nTotWin = sh.item;

    BPROF(1)
    /* ------------------------------ phase T: the totals ------------------------------ */
#pragma unroll 1
    for (int k = threadIdx.x; k < 2 * nTotWin; k += SY_P) {
        const WinTotal w = wtot[k >> 1];
        const int f = k & 1;
        double acc = CP_NEG_INF;
        if (f == 0 || w.second) {
            const int lo = f ? w.nxmin : w.xmin, hi = f ? w.nxmax : w.xmax;
            const double *src = vw + ((long long) (k >> 1) * 2 + f) * SY_P;
            double v[8], nv[8]; /* the next eight terms are in flight while these eight are folded */
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = lo + j <= hi ? src[SY_SLOT(lo + j)] : CP_NEG_INF;
#pragma unroll 1
            for (int x0 = lo; x0 <= hi; x0 += 8) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    nv[j] = x0 + 8 + j <= hi ? src[SY_SLOT(x0 + 8 + j)] : CP_NEG_INF;
#pragma unroll
                for (int j = 0; j < 8; j++) acc = ladd(acc, v[j], cf); /* dpDiagonal_dotProduct :587-597 */
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = nv[j];
            }
        }
        sh.vbuf[threadIdx.x] = acc;
        __builtin_amdgcn_wave_barrier();
        if (f == 0) {
            double tot = acc;
            if (w.second) tot = ladd(acc, sh.vbuf[threadIdx.x + 1], cf);
            wtot[k >> 1].total = tot;
            if (!(fabs(tot - totEst) <= SY_CAND_SLACK)) sh.scan = 1; /* also catches NaN and infinities */
            const long long o = out.nTot + (k >> 1);
            if (o < out.totCap) {
                out.totXay[o] = w.t;
                out.totVal[o] = tot;
            }
        }
    }
    out.nTot += nTotWin;
    __syncthreads();

    BPROF(2)
    /* ------------------------------ phase D: the aligned pairs ------------------------------ */
    /*
     * Two passes over the window's diagonals; in both, every wave walks ALL diagonals but looks only
     * at its own 64 slots (the sweep's mapping), SY_DECODE_U diagonals per iteration with their loads
     * issued together, so HBM latency is paid once per batch instead of once per diagonal.
     *   pass 0  marks the hits: one 64-bit lane mask per (diagonal, wave) to scratch;
     *   prefix  hits per diagonal -> exclusive offsets in emission order (diagonals descending);
     *   pass 1  re-reads only the hit lanes, ranks each hit inside its diagonal by k-mer index
     *           (the reference's x-y order) from the four masks, and writes the triples.
     */
    if (P.mode == 0 && nPost > 0) {
        int *const off = cntBuf;
        auto lane64 = [&](const unsigned long long v, const int src) __attribute__((always_inline)) {
            const unsigned lo = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) v, src);
            const unsigned hi = (unsigned) __builtin_amdgcn_readlane((int) (unsigned) (v >> 32), src);
            return ((unsigned long long) hi << 32) | lo;
        };
        /* masks are written by one wave (or by atomics) and read by the others: read them at agent
         * scope so that a line cached by this CU earlier cannot be served stale */
        auto ld_msk = [&](const long long i) __attribute__((always_inline)) {
            return __hip_atomic_load(msk + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto hits_of = [&](const int k) __attribute__((always_inline)) {
            int n = 0; /* masks are kept four to a diagonal whatever SY_R is */
#pragma unroll
            for (int w2 = 0; w2 < SY_R; w2++) n += __popcll(ld_msk(k * 4ll + w2));
            return n;
        };
        auto prefix = [&]() __attribute__((always_inline)) {
                /* hits per diagonal from the masks; exclusive prefix in emission order */
                int *part = (int *) sh.wbuf;
                int carry = 0; /* hits of the rounds before this one */
#pragma unroll 1
                for (int base = 0; base < nPost; base += 8 * SY_P) { /* eight diagonals per thread and round */
                    const int b0 = base + (int) threadIdx.x * 8;
                    int h[8];
                    int sum = 0;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        h[j] = b0 + j < nPost ? hits_of(b0 + j) : 0;
                        sum += h[j];
                    }
                    /* exclusive scan over the 256 threads: shuffles inside a wave, LDS across the four */
                    int inc = sum;
#pragma unroll
                    for (int o2 = 1; o2 < 64; o2 <<= 1) {
                        const int up = __shfl_up(inc, o2);
                        if (lane >= o2) inc += up;
                    }
                    __syncthreads(); /* part[] of the previous round has been read */
                    if (lane == 63) part[wave] = inc;
                    __syncthreads();
                    int o = carry + inc - sum;
#pragma unroll
                    for (int w2 = 0; w2 < SY_R; w2++) {
                        const int c = part[w2];
                        if (w2 < wave) o += c;
                    }
#pragma unroll
                    for (int w2 = 0; w2 < SY_R; w2++) carry += part[w2];
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (b0 + j < nPost) {
                            off[b0 + j] = o;
                            o += h[j];
                        }
                }
                if (threadIdx.x == 0) sh.cnt[0][0][0] = carry;
                __syncthreads();
                BPROF(5)
        };
        const bool scan = P.scanDecode != 0 || sh.scan != 0;
        if (!scan) {
            /*
             * Decode from the sweep's candidate lists (a few cells per diagonal instead of the whole
             * band): every wave walks its own list twice.  First the exact test of
             * diagonalCalculationPosteriorMatchProbs marks the hits in the (diagonal, wave) masks, then,
             * with the per-diagonal offsets known, each hit is ranked inside its diagonal and written.
             */
            const int nC = sh.cnt[1][wave][0];
            for (int i = threadIdx.x; i < nPost * 4; i += SY_P) msk[i] = 0ull;
            __syncthreads();
            for (int pass = 0; pass < 2; pass++) {
#pragma unroll 1
                for (int i = lane; i < nC; i += 64) {
                    const int2 kx = myKx[i];
                    const int k = kx.x, x = kx.y, t = tPost0 - k;
                    const double ee = myFb[i] - wtot[k / 10].total;
                    if (!(x >= 1 && x <= t - 1 && ee >= P.logThrSlack)) continue;
                    double p = exp(ee);
                    if (!(p >= P.threshold)) continue;
                    if (!pass) {
                        atomicOr(msk + k * 4ll + wave, 1ull << (x & 63));
                        continue;
                    }
                    /* rank inside the diagonal, as in the scan below */
                    const int xmin = bandTab[t].x;
                    const int W0 = SY_WAVE_OF(xmin), s0 = xmin & 63, c = SY_WMOD(wave - W0);
                    const unsigned long long fromS0 = ~0ull << s0, below = (1ull << lane) - 1ull;
                    const unsigned long long own = ld_msk(k * 4ll + wave);
                    int first = 0, beforeMine = 0, others = 0;
#pragma unroll
                    for (int w2 = 0; w2 < SY_R; w2++) {
                        const unsigned long long m2 = ld_msk(k * 4ll + w2);
                        const int c2 = SY_WMOD(w2 - W0);
                        if (c2 == 0) first = __popcll(m2 & fromS0);
                        else {
                            others += __popcll(m2);
                            if (c2 < c) beforeMine += __popcll(m2);
                        }
                    }
                    const int xl = x & 63;
                    const unsigned long long belowX = (1ull << xl) - 1ull;
                    int rank;
                    if (c != 0) rank = first + beforeMine + __popcll(own & belowX);
                    else if (xl >= s0) rank = __popcll(own & fromS0 & belowX);
                    else rank = first + others + __popcll(own & belowX);
                    const long long idx = out.nPairs + off[k] + rank;
                    if (idx < out.pairCap) {
                        if (p > 1.0) p = 1.0;
                        long long *o = out.pairs + idx * 3;
                        o[0] = (long long) floor(p * 10000000.0);
                        o[1] = x - 1;
                        o[2] = t - x - 1;
                        out.logp[idx] = ee;
                    }
                    (void) below;
                }
                if (!pass) { BPROF(3) } else { BPROF(6) }
                __syncthreads();
                if (!pass) { BPROF(4) } else { BPROF(7) }
                if (!pass) prefix();
            }
        } else
        for (int pass = 0; pass < 2; pass++) {
            int xs = wave * 64 + lane; /* this slot's k-mer: the one in (xmax-P, xmax] */
            {
                int a, b;
                band_load(bandTab, tPost0, a, b);
                xs += ((a - xs + SY_P - 1) / SY_P) * SY_P;
                if (xs > b) xs -= SY_P;
            }
            /* pass 1 works from the masks and offsets of a batch, fetched one batch ahead, one
             * (diagonal, wave) mask per lane 0..4U-1 and one offset per lane 0..U-1 */
            unsigned long long mvNext = 0ull;
            int ovNext = 0;
            /* the band of a batch's diagonals: one entry per lane 0..U-1, also a batch ahead */
            int2 bvNext = make_int2(1, 0);
            if (lane < SY_DECODE_U && lane < nPost) bvNext = bandTab[tPost0 - lane];
            if (pass) {
                const int kk = lane >> 2;
                if (lane < 4 * SY_DECODE_U && kk < nPost) mvNext = ld_msk(kk * 4ll + (lane & 3));
                if (lane < SY_DECODE_U && lane < nPost) ovNext = off[lane];
            }
#pragma unroll 1
            for (int k0 = 0; k0 < nPost; k0 += SY_DECODE_U) {
                const unsigned long long mv = mvNext;
                const int ov = ovNext;
                const int2 bv = bvNext;
                int xminA[SY_DECODE_U], xmaxA[SY_DECODE_U], xsA[SY_DECODE_U];
                double e[SY_DECODE_U], tot[SY_DECODE_U];
                unsigned long long own[SY_DECODE_U];
#pragma unroll
                for (int j = 0; j < SY_DECODE_U; j++) {
                    const int k = k0 + j, t = tPost0 - k;
                    const bool live = k < nPost;
                    xminA[j] = __builtin_amdgcn_readlane(bv.x, j);
                    xmaxA[j] = __builtin_amdgcn_readlane(bv.y, j);
                    if (live && xs > xmaxA[j]) xs -= SY_P;
                    xsA[j] = xs;
                    bool want = live && xs >= xminA[j];
                    own[j] = 0ull;
                    if (pass) {
                        own[j] = lane64(mv, j * 4 + wave);
                        want = ((own[j] >> lane) & 1ull) != 0ull;
                    }
                    /* F.match + B.match, parked by the sweep in ring slot 3 */
                    const double *p = want ? g.rp(t, 3) : g.rw;
                    e[j] = *p;
                    tot[j] = wtot[(live ? k : 0) / 10].total;
                }
                {
                    const int ko = k0 + SY_DECODE_U + lane;
                    bvNext = make_int2(1, 0);
                    if (lane < SY_DECODE_U && ko < nPost) bvNext = bandTab[tPost0 - ko];
                }
                if (pass) {
                    const int kk = k0 + SY_DECODE_U + (lane >> 2), ko = k0 + SY_DECODE_U + lane;
                    mvNext = 0ull;
                    ovNext = 0;
                    if (lane < 4 * SY_DECODE_U && kk < nPost) mvNext = ld_msk(kk * 4ll + (lane & 3));
                    if (lane < SY_DECODE_U && ko < nPost) ovNext = off[ko];
                }
#pragma unroll
                for (int j = 0; j < SY_DECODE_U; j++) {
                    const int k = k0 + j, t = tPost0 - k, x = xsA[j];
                    const double ee = e[j] - tot[j];
                    if (!pass) {
                        if (k < nPost) {
                            const int xlo = xminA[j] > 1 ? xminA[j] : 1, xhi = xmaxA[j] < t - 1 ? xmaxA[j] : t - 1;
                            const bool ok = x >= xlo && x <= xhi && ee >= P.logThrSlack;
                            unsigned long long m = 0ull;
                            if (__ballot(ok) != 0ull) {
                                bool hit = false;
                                if (ok) hit = exp(ee) >= P.threshold;
                                m = __ballot(hit);
                            }
                            if (lane == 0) msk[k * 4 + wave] = m;
                        }
                    } else if (own[j] != 0ull) {
                        /* hits of this diagonal with a smaller k-mer index: the band starts in wave W0 at
                         * lane s0 and wraps around the four waves, possibly back into W0's low lanes */
                        const int W0 = SY_WAVE_OF(xminA[j]), s0 = xminA[j] & 63;
                        const unsigned long long fromS0 = ~0ull << s0;
                        const int c = SY_WMOD(wave - W0);
                        int first = 0, beforeMine = 0, others = 0;
#pragma unroll
                        for (int w2 = 0; w2 < SY_R; w2++) {
                            const unsigned long long m2 = lane64(mv, j * 4 + w2);
                            const int c2 = SY_WMOD(w2 - W0);
                            if (c2 == 0) first = __popcll(m2 & fromS0);
                            else {
                                others += __popcll(m2);
                                if (c2 < c) beforeMine += __popcll(m2);
                            }
                        }
                        const unsigned long long below = (1ull << lane) - 1ull;
                        int rank;
                        if (c != 0) rank = first + beforeMine + __popcll(own[j] & below);
                        else if (lane >= s0) rank = __popcll(own[j] & fromS0 & below);
                        else rank = first + others + __popcll(own[j] & below);
                        const long long idx = out.nPairs + __builtin_amdgcn_readlane(ov, j) + rank;
                        if (((own[j] >> lane) & 1ull) != 0ull && idx < out.pairCap) {
                            double p = exp(ee);
                            if (p > 1.0) p = 1.0;
                            long long *o = out.pairs + idx * 3;
                            o[0] = (long long) floor(p * 10000000.0);
                            o[1] = x - 1;
                            o[2] = t - x - 1;
                            out.logp[idx] = ee;
                        }
                    }
                }
            }
            if (!pass) { BPROF(3) } else { BPROF(6) }
            __syncthreads();
            if (!pass) { BPROF(4) } else { BPROF(7) }
            if (!pass) prefix();
        }
        out.nPairs += sh.cnt[0][0][0];
    }
    BPROF(8)
    BPROF_FLUSH
}

} // namespace

/* per-alignment scratch: [hit offsets | window totals | their terms | hit masks | candidate lists] */
static __host__ __device__ long long scratch_cand_offset(int ringD) {
    return 2ll * ringD * sizeof(int) + ((long long) ringD / 10 + 8) * (sizeof(WinTotal) + 2 * SY_P * sizeof(double))
           + 4ll * ringD * sizeof(unsigned long long);
}

/* One workgroup per alignment: forward sweep up to its next traceback point. */
extern "C" __global__ __launch_bounds__(SY_P) void SY_SYM(cpecan_k_sy_forward)(
    const DevItem *__restrict__ items, long long nItems, DevParams P,
    const int2 *__restrict__ bandTab, const double *__restrict__ track,
    const long long *__restrict__ trackBase, const double *__restrict__ events,
    const double *__restrict__ models, double *Fring, long long ringDoubles, int ringD,
    SyState *states) {
    __shared__ Shared sh;
    __shared__ Feed fd;
    __shared__ BandFeed bf;
    const long long idx = blockIdx.x;
    if (idx >= nItems) return;
    SyState *state = states + idx;
    const DevItem it = uniform_item(items[idx]);
    if (state->finished || it.lX + it.lY == 0) return;
    init_coef(sh.coef);
    __syncthreads();
    forward_window(it, P, bandTab + it.diagBase, track + trackBase[idx] * CP_ROW, events,
                   models + (long long) it.model * CP_MODEL_STRIDE, Fring + idx * ringDoubles, ringD,
                   state, sh, fd, bf);
}

/* One workgroup per alignment: backward sweep + posterior decode of the window just described. */
extern "C" __global__ __launch_bounds__(SY_P) SY_BACKWARD_ATTR void SY_SYM(cpecan_k_sy_backward)(
    const DevItem *__restrict__ items, long long nItems, DevParams P,
    const int2 *__restrict__ bandTab, const double *__restrict__ track,
    const long long *__restrict__ trackBase, const double *__restrict__ models, double *Fring,
    long long ringDoubles, int ringD, SyState *states, long long *pairs, double *pairLogp,
    long long *totXay, double *totVal, char *scratch, long long scratchBytes, double *Bring, int window) {
    __shared__ Shared sh;
    __shared__ BandFeed bf;
    const long long idx = blockIdx.x;
    if (idx >= nItems) return;
    SyState *state = states + idx;
    if (!state->winValid) return;
    const DevItem it = uniform_item(items[idx]);
    init_coef(sh.coef);
    __syncthreads();
    ItemOut out;
    out.pairs = pairs + it.pairBase * 3;
    out.logp = pairLogp + it.pairBase;
    out.pairCap = it.pairCap;
    out.totXay = totXay + it.totBase;
    out.totVal = totVal + it.totBase;
    out.totCap = it.totCap;
    out.nPairs = uni64(state->nPairs);
    out.nTot = uni64(state->nTot);
    backward_window(it, P, bandTab + it.diagBase, track + trackBase[idx] * CP_ROW,
                    models + (long long) it.model * CP_MODEL_STRIDE, Fring + idx * ringDoubles, ringD,
                    state, out, sh, bf, (int *) (scratch + idx * scratchBytes),
                    (WinTotal *) (scratch + idx * scratchBytes + 2ll * ringD * sizeof(int)),
                    (double *) (scratch + idx * scratchBytes + 2ll * ringD * sizeof(int)
                                + ((long long) ringD / 10 + 8) * sizeof(WinTotal)),
                    (unsigned long long *) (scratch + idx * scratchBytes + 2ll * ringD * sizeof(int)
                                            + ((long long) ringD / 10 + 8)
                                                  * (sizeof(WinTotal) + 2 * SY_P * sizeof(double))),
                    (int2 *) (scratch + idx * scratchBytes + scratch_cand_offset(ringD)),
                    (double *) (scratch + idx * scratchBytes + scratch_cand_offset(ringD)
                                + (long long) SY_R * SY_CAND_PER_DIAG * ringD * sizeof(int2)),
                    Bring ? Bring + idx * ((long long) ringD * SY_R * 3 * 64) : nullptr);
    if (threadIdx.x == 0) {
        state->nPairs = out.nPairs;
        state->nTot = out.nTot;
        state->winValid = 0;
        state->expectPending = P.mode != 0 ? window + 1 : 0; /* which launch's window the B ring holds */
    }
}

/*
 * Baum-Welch expectations of the traceback window the backward kernel just swept
 * (diagonalCalculation_Expectations :841-863 with cell_signal_updateTransAndKmerSkipExpectations
 * :426-443).  By now every operand is in HBM -- forward cells and the two event-dependent emissions in
 * the forward ring, backward cells in the B ring, the window's exact totals in scratch -- so this is
 * an element-wise pass with no recurrence and no barrier: per cell eight exp(F.from + B.to + (eP + tP)
 * - total), summed per thread and reduced once per window.  A lane keeps the sum of its k-mer's
 * gap-X expectations in a register and adds it to the k-mer's bin when its slot moves to another k-mer.
 * The match block is skipped on the window's two lowest diagonals' worth of reach, as in the
 * reference, where forward[t-2] has been freed by then (quirk kept by the general kernel too).
 */
extern "C" __global__ __launch_bounds__(SY_P) void SY_SYM(cpecan_k_sy_expect)(
    const DevItem *__restrict__ items, long long nItems, DevParams P, const int2 *__restrict__ bandTab,
    const double *__restrict__ track, const long long *__restrict__ trackBase,
    const unsigned short *__restrict__ kidx, const double *__restrict__ models, const double *Fring,
    long long ringDoubles, const double *Bring, int ringD, SyState *states, const char *scratch,
    long long scratchBytes, double *expect, int window) {
    __shared__ double sExp[16];
    const long long idx = blockIdx.x;
    if (idx >= nItems) return;
    const SyState *state = states + idx;
    /* the pass has no recurrence: gridDim.y workgroups share a window's diagonals, a contiguous run each; the
     * state record is only read here (the backward kernel stamps it with the launch it belongs to) */
    if (state->expectPending != window + 1) return;
    const DevItem it = uniform_item(items[idx]);
    const double *model = models + (long long) it.model * CP_MODEL_STRIDE;
    double T[9];
#pragma unroll
    for (int i = 0; i < 9; i++) T[i] = model[i];
    const Geometry g = make_geometry(const_cast<double *>(Fring) + idx * ringDoubles, ringD);
    const int lane = g.lane, wave = g.wave;
    const double *bown = Bring + idx * ((long long) ringD * SY_R * 3 * 64) + wave * (3 * 64) + lane;
    const double *tr = track + trackBase[idx] * CP_ROW;
    const unsigned short *kx = kidx + it.xOff;
    const int2 *tab = bandTab + it.diagBase;
    const WinTotal *wtot = (const WinTotal *) (scratch + idx * scratchBytes + 2ll * ringD * sizeof(int));
    const int dTop = uni(state->winTop), from = uni(state->winFrom), to = uni(state->winTo);
    const int tPost0 = dTop < from ? dTop : from;
    double *dst = expect + (long long) it.model * (9 + 4096 + 1);

    double acc[8]; /* M>X X>X Y>X | M>M X>M Y>M | M>Y Y>Y */
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = 0.0;
    double lik = 0.0, gapSum = 0.0;
    int gapX = -1; /* matrix column whose gap-X expectations gapSum holds */

    const int perChunk = (tPost0 - to + (int) gridDim.y - 1) / (int) gridDim.y;
    const int tHi = tPost0 - (int) blockIdx.y * perChunk;             /* this workgroup: diagonals tHi .. tLo+1 */
    const int tLo = tHi - perChunk > to ? tHi - perChunk : to;
    if (tHi <= to) return;
    int xs = wave * 64 + lane; /* this slot's k-mer: the one in (xmax-P, xmax] */
    int b0min, b0max, b1min, b1max, b2min, b2max;
    band_load(tab, tHi, b0min, b0max);
    band_load(tab, tHi - 1, b1min, b1max);
    xs += ((b0min - xs + SY_P - 1) / SY_P) * SY_P;
    if (xs > b0max) xs -= SY_P;
#pragma unroll 1
    for (int t = tHi; t > tLo; t--) {
        band_load(tab, t - 2, b2min, b2max);
        if (xs > b0max) xs -= SY_P;
        const int x = xs;
        const double total = wtot[(tPost0 - t) / 10].total;
        if (threadIdx.x == 0) lik += total;
        if (x >= b0min) { /* the cell (t, x) exists */
            const double *bc = bown + (long long) (t & g.ringMask) * (SY_R * 3 * 64);
            const double Bm = bc[0], Bx = bc[64], By = bc[128];
            const bool vLower = x - 1 >= b1min && x - 1 <= b1max;
            const bool vMiddle = t - 2 >= to && x - 1 >= b2min && x - 1 <= b2max;
            const bool vUpper = x >= b1min && x <= b1max;
            if (vLower) {
                const double l0 = *g.rpb(t - 1, 0), l1 = *g.rpb(t - 1, 1), l2 = *g.rpb(t - 1, 2);
                const double eP = tr[(long long) x * CP_ROW + CP_GAPX];
                const double p0 = exp(l0 + Bx + (eP + T[T_GAP_OPEN_X]) - total);
                const double p1 = exp(l1 + Bx + (eP + T[T_GAP_EXTEND_X]) - total);
                const double p2 = exp(l2 + Bx + (eP + T[T_GAP_SWITCH_TO_X]) - total);
                acc[0] += p0;
                acc[1] += p1;
                acc[2] += p2;
                if (x != gapX) {
                    if (gapX > 0) {
                        const int k = kx[gapX - 1];
                        if (k < 4096) atomicAdd(dst + 9 + k, gapSum);
                    }
                    gapX = x;
                    gapSum = 0.0;
                }
                gapSum += p0;
                gapSum += p1;
                gapSum += p2;
            }
            if (vMiddle) {
                const double m0 = *g.rpb(t - 2, 0), m1 = *g.rpb(t - 2, 1), m2 = *g.rpb(t - 2, 2);
                const double eP = *g.rp(t, 3);
                acc[3] += exp(m0 + Bm + (eP + T[T_MATCH_CONTINUE]) - total);
                acc[4] += exp(m1 + Bm + (eP + T[T_MATCH_FROM_GAP_X]) - total);
                acc[5] += exp(m2 + Bm + (eP + T[T_MATCH_FROM_GAP_Y]) - total);
            }
            if (vUpper) {
                const double u0 = *g.rp(t - 1, 0), u2 = *g.rp(t - 1, 2);
                const double eP = *g.rp(t, 4);
                acc[6] += exp(u0 + By + (eP + T[T_GAP_OPEN_Y]) - total);
                acc[7] += exp(u2 + By + (eP + T[T_GAP_EXTEND_Y]) - total);
            }
        }
        b0min = b1min; b0max = b1max;
        b1min = b2min; b1max = b2max;
    }
    if (gapX > 0) {
        const int k = kx[gapX - 1];
        if (k < 4096) atomicAdd(dst + 9 + k, gapSum);
    }
    /* block reduction of the per-thread sums, then one atomic per value */
    if (threadIdx.x < 16) sExp[threadIdx.x] = 0.0;
    __syncthreads();
    const int slot[8] = { 0 * 3 + 1, 1 * 3 + 1, 2 * 3 + 1, 0 * 3 + 0, 1 * 3 + 0, 2 * 3 + 0, 0 * 3 + 2, 2 * 3 + 2 };
#pragma unroll
    for (int i = 0; i < 8; i++) {
        double v = acc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
        if (lane == 0) atomicAdd(&sExp[slot[i]], v);
    }
    __syncthreads();
    if (threadIdx.x < 9) atomicAdd(dst + threadIdx.x, sExp[threadIdx.x]);
    if (threadIdx.x == 0) atomicAdd(dst + 9 + 4096, lik);
}

#if SY_R == 4
/* results of the per-alignment states into the batch's count arrays */
extern "C" __global__ void cpecan_k_sy_counts(const SyState *states, long long nItems,
                                              long long *nPairs, long long *nTot, long long *nCells) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nItems) return;
    nPairs[i] = states[i].nPairs;
    nTot[i] = states[i].nTot;
    nCells[i] = states[i].cells;
}

/* per-item track of emission constants: row x (0..lX) = model row of the k-mer that matrix column x
 * scores (column 0 = the "not a k-mer" sentinel, sequence_getKmer index -1, :314-318) */
extern "C" __global__ void cpecan_k_track(const DevItem *__restrict__ items, long long nItems,
                                          const long long *__restrict__ trackBase,
                                          const unsigned short *__restrict__ kidx,
                                          const double *__restrict__ models, double *track) {
    const long long item = blockIdx.y;
    if (item >= nItems) return;
    const DevItem it = items[item];
    const double *rows = models + (long long) it.model * CP_MODEL_STRIDE + CP_MODEL_HEADER;
    const long long n = (it.lX + 1) * CP_ROW;
    double *dst = track + trackBase[item] * CP_ROW;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long) gridDim.x * blockDim.x) {
        const long long x = i / CP_ROW;
        const int j = (int) (i - x * CP_ROW);
        const int k = x == 0 ? 4096 : (int) kidx[it.xOff + x - 1];
        dst[i] = rows[(long long) k * CP_ROW + j];
    }
}

/* division self-test (see cpecan_hip_selftest_division) */
extern "C" __global__ void cpecan_k_divtest(long long n, unsigned long long seed, unsigned long long *bad) {
    long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long local = 0;
    for (; i < n; i += (long long) gridDim.x * blockDim.x) {
        unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long) (i + 1);
        double u[3];
        for (int k = 0; k < 3; k++) { /* splitmix64 */
            z += 0x9E3779B97F4A7C15ull;
            unsigned long long r = z;
            r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull;
            r = (r ^ (r >> 27)) * 0x94D049BB133111EBull;
            r ^= r >> 31;
            u[k] = (double) (r >> 11) * (1.0 / 9007199254740992.0);
        }
        const bool noise = (i & 1) != 0;
        const double x = noise ? 0.001 + 4.0 * u[0] : 30.0 + 70.0 * u[0];
        const double mu = noise ? 0.3 + 2.0 * u[1] : 40.0 + 45.0 * u[1];
        const double sd = noise ? 0.05 + 1.5 * u[2] : 0.3 + 4.0 * u[2];
        const double rsd = 1.0 / sd;
        const double t = x - mu;
        const double q = t * rsd;
        const double rem = __fma_rn(-q, sd, t);
        const double a = __fma_rn(rem, rsd, q);
        const double ref = t / sd;
        if (!(a == ref)) local++;
    }
    if (local) atomicAdd(bad, local);
}

extern "C" int cpecan_systolic_divtest(hipStream_t stream, long long n, unsigned long long seed,
                                       unsigned long long *bad) {
    hipLaunchKernelGGL(cpecan_k_divtest, dim3(1024), dim3(256), 0, stream, n, seed, bad);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif /* SY_R == 4 */

#ifdef SY_PROFILE
extern "C" int SY_SYM(cpecan_systolic_prof_fetch)(unsigned long long *dst) {
    unsigned long long zero[80] = {0};
    if (hipMemcpyFromSymbol(dst, HIP_SYMBOL(sy_prof), sizeof(zero)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(sy_prof), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#endif
extern "C" int SY_SYM(cpecan_systolic_max_width)(void) { return SY_P - 2 * SY_PREFETCH; }
extern "C" int SY_SYM(cpecan_systolic_rows)(void) { return SY_R; }
extern "C" int SY_SYM(cpecan_systolic_ring_row_doubles)(void) { return SY_R * SY_RING_VALUES * 64; }

extern "C" int SY_SYM(cpecan_systolic_bring_row_doubles)(void) { return SY_R * 3 * 64; }
#if SY_R == 4
extern "C" int cpecan_systolic_state_bytes(void) { return (int) sizeof(SyState); }
#endif
/* HBM scratch per alignment: a hit count and an output offset per ring diagonal, and per refresh of the window one
 * WinTotal and the two rows of per-cell terms */
extern "C" long long SY_SYM(cpecan_systolic_scratch_bytes)(int ringD) {
    return scratch_cand_offset(ringD)
           + (long long) SY_R * SY_CAND_PER_DIAG * ringD * (sizeof(int2) + sizeof(double));
}

/* Launchers of the four stages of one pass over a batch (the C-ABI layer sequences them:
 * track, then per window {forward, backward}, then counts). */
#if SY_R == 4
extern "C" int cpecan_systolic_launch_track(hipStream_t stream, const DevItem *items, long long nItems,
                                            const double *track, const long long *trackBase,
                                            const unsigned short *kidx, const double *models,
                                            void *states, int maxLX) {
    int bx = (int) ((((long long) maxLX + 1) * CP_ROW + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(cpecan_k_track, dim3(bx, (unsigned) nItems), dim3(256), 0, stream, items,
                       nItems, trackBase, kidx, models, (double *) track);
    if (hipMemsetAsync(states, 0, (size_t) nItems * sizeof(SyState), stream) != hipSuccess) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif
extern "C" int SY_SYM(cpecan_systolic_launch_forward)(hipStream_t stream, const DevItem *items, long long nItems,
                                              DevParams P, const void *bandTab, const double *track,
                                              const long long *trackBase, const double *events,
                                              const double *models, double *Fring,
                                              long long ringDoubles, int ringD, void *states) {
    hipLaunchKernelGGL(SY_SYM(cpecan_k_sy_forward), dim3((unsigned) nItems), dim3(SY_P), 0, stream, items, nItems,
                       P, (const int2 *) bandTab, track, trackBase, events, models, Fring, ringDoubles, ringD,
                       (SyState *) states);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int SY_SYM(cpecan_systolic_launch_backward)(hipStream_t stream, const DevItem *items, long long nItems,
                                               DevParams P, const void *bandTab, const double *track,
                                               const long long *trackBase, const double *models,
                                               double *Fring, long long ringDoubles, int ringD,
                                               void *states, long long *pairs, double *pairLogp,
                                               long long *totXay, double *totVal, char *scratch,
                                               long long scratchBytes, double *Bring, int window) {
    hipLaunchKernelGGL(SY_SYM(cpecan_k_sy_backward), dim3((unsigned) nItems), dim3(SY_P), 0, stream, items, nItems,
                       P, (const int2 *) bandTab, track, trackBase, models, Fring, ringDoubles, ringD,
                       (SyState *) states, pairs, pairLogp, totXay, totVal, scratch, scratchBytes, Bring, window);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int SY_SYM(cpecan_systolic_launch_expect)(hipStream_t stream, const DevItem *items, long long nItems,
                                             DevParams P, const void *bandTab, const double *track,
                                             const long long *trackBase, const unsigned short *kidx,
                                             const double *models, const double *Fring,
                                             long long ringDoubles, const double *Bring, int ringD,
                                             void *states, const char *scratch, long long scratchBytes,
                                             double *expect, int window, long long *pairs, double *pairLogp) {
    (void) pairs; (void) pairLogp; /* (the HDP machine's wave kernels append event assignments there) */
    hipLaunchKernelGGL(SY_SYM(cpecan_k_sy_expect), dim3((unsigned) nItems, SY_EXPECT_CHUNKS), dim3(SY_P), 0, stream,
                       items, nItems, P, (const int2 *) bandTab, track, trackBase, kidx, models, Fring, ringDoubles,
                       Bring, ringD, (SyState *) states, scratch, scratchBytes, expect, window);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#if SY_R == 4
extern "C" int cpecan_systolic_launch_counts(hipStream_t stream, const void *states, long long nItems,
                                             long long *nPairs, long long *nTot, long long *nCells) {
    hipLaunchKernelGGL(cpecan_k_sy_counts, dim3((unsigned) ((nItems + 255) / 256)), dim3(256), 0, stream,
                       (const SyState *) states, nItems, nPairs, nTot, nCells);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif

extern "C" int SY_SYM(cpecan_systolic_occupancy)(int *workgroupsPerCU) {
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, SY_SYM(cpecan_k_sy_backward), SY_P, 0);
    if (e != hipSuccess) return -1;
    *workgroupsPerCU = n;
    return 0;
}

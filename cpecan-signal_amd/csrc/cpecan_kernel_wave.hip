/*
 * cpecan_kernel_wave.hip -- the throughput kernels: banded forward / backward / posterior DP of the
 * 3-state strawMan signal machine, ONE WAVE PER ALIGNMENT, several cells per lane.
 *
 * Mapping (designed for CDNA4's 64-lane waves):
 *   - an alignment is swept by a single wave; the band's k-mers live in P = 64 * L register slots,
 *     slot s = x mod P held by lane s / L as its cell number s % L ("layer"): a lane owns L
 *     consecutive k-mers, with their emission constants resident in VGPRs while they are in the band.
 *   - one loop iteration = one anti-diagonal = L independent cell bodies per lane.  A cell's
 *     neighbour (x-1, .) is the same lane's previous layer, a register, except for layer 0, which takes
 *     layer L-1 of the lane below: ONE wave rotation (DPP wave_ror / wave_rol) of three values per
 *     diagonal, whatever L is.  No LDS exchange, no barrier, no second wave to wait for.
 *   - cells outside the band cost no select: a slot whose k-mer is not in the band holds the "not a
 *     k-mer" constants (every emission -inf), so its cells come out -inf by themselves and nothing it
 *     holds can reach a cell of the band (forward: parked rows; backward: parked gap-X sums and a dummy
 *     ring row of -inf emissions for the lanes that have no cell on a diagonal).
 *   - band edges move by at most one k-mer per diagonal; the launch's edge steps sit in LDS as two
 *     bit strings, so entering / leaving k-mers are found on the scalar unit.
 *   - forward cells go to HBM once into a per-alignment ring of diagonals
 *     ([diagonal][layer][(Fm,pm) x 64 | py x 64 | (Fx,Fy) x 64]) and are read once by the sweep back, three diagonals
 *     ahead of use; the forward sweep's inputs (events, k-mer rows) are staged in LDS per block of
 *     diagonals, so neither loop waits for a load it has just issued.
 *   - two kernels per traceback window, sequenced by the C-ABI layer (cpecan_hip.hip); WvState and the
 *     ring carry over.  The posterior decode works from candidate lists the sweep back collects; a
 *     window whose candidates cannot be trusted is swept again with the exact totals in hand.
 * MFMA is not used: the recurrence is a scan with an approximate log-add, not a contraction.
 *
 * Numerics: identical to the general kernel and the CPU oracle, bit for bit (cpecan_device.h; the
 * division (x - mu) / sigma is a Markstein-corrected multiply by the host-rounded reciprocal).
 *
 * Reference: getPosteriorProbsWithBanding impl/pairwiseAligner.c:870-1006, diagonalCalculation* :681-863,
 * stateMachine3_cellCalculate impl/stateMachine.c:1305-1334, logAdd impl/pairwiseAligner.c:238-255.
 */
#include "cpecan_device.h"
#include "cpecan_sweep.h"

#include <vector>

#ifndef WV_L
#define WV_L 3 /* cells per lane: 1..4 (bands up to 56, 120, 184, 248 k-mers) */
#endif
#define WV_P (64 * WV_L)
/* -DWV_HDP: the same sweeps for the 3-state HDP signal machine (stateMachine3HDP_cellCalculate, impl/stateMachine.c:
 * 1338-1370): one density of the NanoporeHDP serves as match and as gap-Y emission, the gap-X emission is a flat
 * log(0.1); symbols suffixed _h2.._h4 */
/* -DWV_VANILLA: the 3-state vanilla signal machine (stateMachine3Vanilla_cellCalculate, impl/stateMachine.c:1368-1409):
 * transition probabilities per reference position (30 skip bins of the k-mer pair sequence_getKmer2 exposes), a
 * Gaussian level term plus an inverse-Gaussian noise term per emission; symbols suffixed _v2.._v4 */
#if defined(WV_VANILLA) && WV_L == 4
#define WV_SYM(n) n##_v4
#elif defined(WV_VANILLA) && WV_L == 3
#define WV_SYM(n) n##_v3
#elif defined(WV_VANILLA)
#define WV_SYM(n) n##_v2
#elif defined(WV_HDP) && WV_L == 4
#define WV_SYM(n) n##_h4
#elif defined(WV_HDP) && WV_L == 3
#define WV_SYM(n) n##_h3
#elif defined(WV_HDP)
#define WV_SYM(n) n##_h2
#elif WV_L == 4
#define WV_SYM(n) n##_l4
#elif WV_L == 3
#define WV_SYM(n) n##_l3
#elif WV_L == 2
#define WV_SYM(n) n##_l2
#else
#define WV_SYM(n) n##_l1
#endif
#if defined(WV_VANILLA)
#define WV_MODEL_DOUBLES ((long long) CP_VMODEL_STRIDE)
#elif defined(WV_HDP)
#define WV_MODEL_DOUBLES ((long long) (sizeof(DevHdpModel) / sizeof(double))) /* a model = one DevHdpModel record */
#else
#define WV_MODEL_DOUBLES ((long long) CP_MODEL_STRIDE)
#endif
#ifdef WV_VANILLA
/* doubles per column of the track: per table (match, extra event) level mu, sd, 1/sd, K and noise mean, 1/mean, lambda,
 * log(lambda) - log(2 pi); then the five log transition probabilities of the column's skip bin: into gap X from match
 * and from gap X, into match from match and from gap X, into gap Y from match */
#define WV_ROW 22
#define WV_PXW 6             /* of which the sweep back keeps the last six (three pairs) per slot */
#else
#define WV_ROW 20            /* doubles per column of the track: 16 emission constants, gap-X sums (open, extend, switch), gap-X */
#define WV_PXW 4
#endif
#define WV_ROWN 32           /* LDS ring of k-mer rows (>= the feed block)                              */
#define WV_FEED_MAX 32       /* diagonals per feed block of the forward sweep                            */
#define WV_BITWORDS 256      /* band edge steps kept in LDS: 32 diagonals per word, circular, re-staged in halves */
/* A ring row of the forward sweep, in doubles: per layer 64 (Fm, pm) pairs | 64 (Fx, Fy) pairs; after the layers the
 * gap-Y emissions, the layers two by two as 64 (py, py) pairs, an odd last layer's as 64 doubles -- the sweeps are bound
 * by HBM traffic (tools/pmc_sweeps.sh), so a row carries nothing but its values; all but the last layer's emissions move
 * in 16-byte accesses (8-byte streaming stores take 2.5 times as long per instruction: tools/ubench_vmem.hip) */
#define WV_LAYER_DOUBLES 256
#define WV_OFF_FM(lane) ((lane) * 2)
#define WV_OFF_PM(lane) ((lane) * 2 + 1)
#define WV_OFF_FX(lane) (128 + (lane) * 2)
#define WV_OFF_FY(lane) (128 + (lane) * 2 + 1)
/* from the row's start */
#define WV_ROW_PY(j, lane) \
    (WV_L * WV_LAYER_DOUBLES + ((j) >> 1) * 128 + (((WV_L & 1) && (j) == WV_L - 1) ? (lane) : (lane) * 2 + ((j) & 1)))
#define WV_ROW_DOUBLES (WV_L * WV_LAYER_DOUBLES + (WV_L / 2) * 128 + (WV_L & 1) * 64)
#define WV_PREFETCH 2        /* diagonals the backward sweep fetches ahead (== its unroll factor) */
#define WV_CAND_SLACK 0.25   /* candidates: cells within this (log units) below the posterior threshold */
#define WV_CAND_PER_DIAG 4   /* candidate capacity, in records per ring diagonal and layer */
#define WV_EXPECT_CHUNKS 8   /* workgroups that share one window's diagonals in the expectation pass */
#ifndef WV_BACKWARD_PRIO
#define WV_BACKWARD_PRIO 2
#endif
#ifndef WV_WPB
#define WV_WPB 1             /* alignments (waves) per workgroup of the sweeps: a workgroup's four waves are placed on the
                                four SIMDs of a CU, which single-wave workgroups are not promised */
#endif

#define WV_FOR_LAYER_1(jv, ...) { constexpr int J = 0; (void) (jv); __VA_ARGS__ }
#define WV_FOR_LAYER_2(jv, ...) if ((jv) == 0) { constexpr int J = 0; __VA_ARGS__ } else { constexpr int J = 1; __VA_ARGS__ }
#define WV_FOR_LAYER_3(jv, ...) if ((jv) == 0) { constexpr int J = 0; __VA_ARGS__ } else if ((jv) == 1) { constexpr int J = 1; __VA_ARGS__ } else { constexpr int J = 2; __VA_ARGS__ }
#define WV_FOR_LAYER_4(jv, ...) if ((jv) == 0) { constexpr int J = 0; __VA_ARGS__ } else if ((jv) == 1) { constexpr int J = 1; __VA_ARGS__ } else if ((jv) == 2) { constexpr int J = 2; __VA_ARGS__ } else { constexpr int J = 3; __VA_ARGS__ }
#if WV_L == 4
#define WV_FOR_LAYER WV_FOR_LAYER_4
#elif WV_L == 3
#define WV_FOR_LAYER WV_FOR_LAYER_3
#elif WV_L == 2
#define WV_FOR_LAYER WV_FOR_LAYER_2
#else
#define WV_FOR_LAYER WV_FOR_LAYER_1
#endif

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) const d2 lds_d2;
typedef lds_d2 *lds_d2p;
#else
typedef const d2 *lds_d2p; /* (the host pass only parses the kernels) */
#define lds_d2p_cast(a) ((lds_d2p) (size_t) (a))
#endif
#ifndef lds_d2p_cast
#define lds_d2p_cast(a) ((lds_d2p) (a))
#endif

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v) {
    const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) v);
    const int hi = __builtin_amdgcn_readfirstlane((int) (v >> 32));
    return ((long long) hi << 32) | lo;
}
__device__ __forceinline__ double uni64_d(double v) {
    return __longlong_as_double(uni64(__double_as_longlong(v)));
}
template <typename T> __device__ __forceinline__ unsigned lds_addr(const T *p) {
    return (unsigned) (size_t) (const __attribute__((address_space(3))) T *) p;
}
/* the work item as wave-uniform (scalar) values */
__device__ __forceinline__ DevItem uniform_item(const DevItem &s) {
    DevItem d;
    d.lX = uni64(s.lX); d.lY = uni64(s.lY); d.xOff = uni64(s.xOff); d.yOff = uni64(s.yOff);
    d.anchorOff = 0; d.nAnchors = 0; d.diagBase = uni64(s.diagBase); d.cellBase = 0;
    d.nCells = 0; d.pairBase = uni64(s.pairBase); d.pairCap = uni64(s.pairCap);
    d.totBase = uni64(s.totBase); d.totCap = uni64(s.totCap); d.bwsBase = 0;
    d.model = uni(s.model); d.raggedL = uni(s.raggedL); d.raggedR = uni(s.raggedR); d.maxWidth = uni(s.maxWidth);
    return d;
}
template <typename V> __device__ __forceinline__ V ld_agent(V *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* d < 7.5 ? r : hi as one compare and two lane selects the compiler cannot turn into a divergent branch
 * around the cubic (it does, given the chance, and the sweep then pays a branch per logAdd) */
__device__ __forceinline__ double sel_below(double d, double r, double hi) {
    int lo_, hi_;
    const double thr = 7.5;
    asm("v_cmp_gt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %6, vcc\n\tv_cndmask_b32 %1, %5, %7, vcc"
        : "=&v"(lo_), "=&v"(hi_)
        : "s"(thr), "v"(d), "v"(__double2loint(hi)), "v"(__double2hiint(hi)), "v"(__double2loint(r)),
          "v"(__double2hiint(r))
        : "vcc");
    return __hiloint2double(hi_, lo_);
}

/* logAdd (impl/pairwiseAligner.c:238-255), branch-free and bit-identical: hi/lo are the operands as the
 * reference's two branches order them; its "smaller operand is -inf" and ">= 7.5" exits both yield hi, and
 * (-inf) - (-inf) = NaN fails d < 7.5 exactly like those exits.  The cubic's four float-literal
 * coefficients come from a 512-byte LDS table indexed by n = ceil(2d) (the pieces' limits 1, 2.5, 4.5 are
 * multiples of 1/2 and 2d is exact, so n decides the piece without a comparison); NaN converts to 0 and
 * d >= 7.5 is clamped: either way the cubic is discarded. */
__device__ __forceinline__ double ladd(double x, double y, unsigned coefAddr) {
    double hi, lo;
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(x), "v"(y));
    const double d = hi - lo;
    int n;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(__builtin_ceil(d + d)));
    n = n < 15 ? n : 15;
    const lds_d2p c = lds_d2p_cast(coefAddr + (unsigned) n * 32u);
    const d2 c32 = c[0], c10 = c[1];
    const double r = ((c32.x * d + c32.y) * d + c10.x) * d + c10.y + lo;
    return sel_below(d, r, hi);
}
/* N independent logAdds, stage by stage, so that their table reads are in flight together; the two halves can be
 * called apart, with the first half of the next batch in between: the sweeps keep two batches in flight */
template <int N> struct LaddPending {
    double hi[N], lo[N], d[N];
    d2 c32[N], c10[N];
};
template <int N> __device__ __forceinline__ void ladd_issue(LaddPending<N> &p, const double (&acc)[N], const double (&y)[N],
                                                            unsigned coefAddr) {
#pragma unroll
    for (int k = 0; k < N; k++) {
        asm("v_max_f64 %0, %1, %2" : "=v"(p.hi[k]) : "v"(acc[k]), "v"(y[k]));
        asm("v_min_f64 %0, %1, %2" : "=v"(p.lo[k]) : "v"(acc[k]), "v"(y[k]));
        p.d[k] = p.hi[k] - p.lo[k];
        int n;
        asm("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(__builtin_ceil(p.d[k] + p.d[k])));
        n = n < 15 ? n : 15;
        const lds_d2p c = lds_d2p_cast(coefAddr + (unsigned) n * 32u);
        p.c32[k] = c[0];
        p.c10[k] = c[1];
    }
}
template <int N> __device__ __forceinline__ void ladd_finish(const LaddPending<N> &p, double (&acc)[N]) {
#pragma unroll
    for (int k = 0; k < N; k++) {
        const double r = ((p.c32[k].x * p.d[k] + p.c32[k].y) * p.d[k] + p.c10[k].x) * p.d[k] + p.c10[k].y + p.lo[k];
        acc[k] = sel_below(p.d[k], r, p.hi[k]);
    }
}
template <int N> __device__ __forceinline__ void laddN(double (&acc)[N], const double (&y)[N], unsigned coefAddr) {
    LaddPending<N> p;
    ladd_issue<N>(p, acc, y, coefAddr);
    ladd_finish<N>(p, acc);
}
__device__ __forceinline__ void init_coef(double *coef) {
    const float t[16] = { -0.009350833524763f, 0.130659527668286f, 0.498799810682272f, 0.693203116424741f,
                          -0.014532321752540f, 0.139942324101744f, 0.495635523139337f, 0.692140569840976f,
                          -0.004605031767994f, 0.063427417320019f, 0.695956496475118f, 0.514272634594009f,
                          -0.000458661602210f, 0.009695946122598f, 0.930734667215156f, 0.168037164329057f };
    const int l = threadIdx.x & 63, n = l >> 2, piece = n <= 2 ? 0 : n <= 5 ? 1 : n <= 9 ? 2 : 3;
    coef[l] = (double) t[piece * 4 + (l & 3)];
}

/* log N(x; mu, sd) = K + (-0.5*a*a), a = (x-mu)/sd (impl/stateMachine.c:333-343); the quotient is
 * q + fma(-q, sd, t) * rsd with q = t*rsd, rsd = RN(1/sd): Markstein's correction step, which rounds to
 * the same double as the division.  sd == 0 rows carry rsd = 0, K = -inf => -inf. */
__device__ __forceinline__ double lgauss(double x, double mu, double sd, double rsd, double K) {
    const double t = x - mu;
    const double q = t * rsd;
    const double rem = __fma_rn(-q, sd, t);
    const double a = __fma_rn(rem, rsd, q);
    return K + (-0.5 * a * a);
}

/* lane i <- lane i-1, lane 0 <- lane 63 (DPP wave_ror:1) */
__device__ __forceinline__ double ror1(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x13C, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x13C, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
/* lane i <- lane i+1, lane 63 <- lane 0 (DPP wave_rol:1) */
__device__ __forceinline__ double rol1(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x134, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x134, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast(double v, int srcLane) { /* srcLane wave-uniform */
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srcLane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srcLane);
    return __hiloint2double(hi, lo);
}

/* logAdd-fold of one value per lane into acc (wave-uniform in and out), lanes in ascending order; visits only
 * the lanes that can change the running value */
__device__ __forceinline__ double wave_fold(double acc, double v, unsigned cf) {
    const int lane = threadIdx.x & 63;
    unsigned long long after = ~0ull;
#pragma unroll 1
    for (;;) {
        const bool eff = ((after >> lane) & 1ull) && (v > CP_NEG_INF) && !(acc - v >= 7.5);
        const unsigned long long m = __ballot(eff);
        if (m == 0ull) break;
        const int first = __ffsll((long long) m) - 1;
        acc = ladd(acc, bcast(v, first), cf);
        after = first >= 63 ? 0ull : (~0ull << (first + 1));
    }
    return acc;
}

/* a slot's k-mer constants, as a track row lays them out */
struct Prm {
    d2 a[WV_ROW / 2]; /* (mu,sd) (rsd,K1) (nmu,nsd) (rnsd,K2) | the same for the gap-Y table | (pxo,pxe) (pxs,px) */
};
/* load of a whole row from LDS into one layer's registers in every lane (start-up) */
__device__ __forceinline__ void load_row_all(Prm &p, unsigned rowAddr) {
    const lds_d2p r = lds_d2p_cast(rowAddr);
#pragma unroll
    for (int k = 0; k < WV_ROW / 2; k++) p.a[k] = r[k]; /* (pairs a build never reads are dropped by the compiler) */
}
/* A k-mer enters (or leaves) the band: its row (or the "not a k-mer" row) is loaded from LDS into the registers
 * of ONE slot -- lane laneMask, layer sel -- and the lane masks of the band are updated.  The layer is a run-time
 * (wave-uniform) value and registers cannot be indexed, so the select sits inside one asm statement: seen from
 * the compiler there is no control flow and no copy of the L x 20 constants around it.  fl[layer] is the lane's own
 * "my slot holds a k-mer of the band" flag: it predicates the ring stores. */
#define WV_SLOT_ASM(MASKOP, FLAGV, BODY, OUTS)                                                                  \
    unsigned long long sv;                                                                                       \
    asm volatile("s_mov_b64 %[sv], exec\n\t"                                                                      \
                 "s_mov_b64 exec, %[m]\n\t" BODY(MASKOP)                                                           \
                 "s_mov_b64 exec, %[sv]\n\t"                                                                      \
                 "s_waitcnt lgkmcnt(0)"                                                                            \
                 : [sv] "=&s"(sv) OUTS                                                                             \
                 : [a] "v"(rowAddr), [m] "s"(laneMask), [sel] "s"(sel), [fv] "s"(FLAGV)                            \
                 : "memory", "scc");
#if defined(WV_VANILLA)
#include "cpecan_wave_slots_vanilla.h"
#elif defined(WV_HDP)
#include "cpecan_wave_slots_hdp.h"
#else
#if WV_L == 1
#define WV_SLOT_BODY9(MASKOP) "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "ds_read_b128 %[a2], %[a] offset:32\n\t" "ds_read_b128 %[a3], %[a] offset:48\n\t" "ds_read_b128 %[a4], %[a] offset:64\n\t" "ds_read_b128 %[a5], %[a] offset:80\n\t" "ds_read_b128 %[a6], %[a] offset:96\n\t" "ds_read_b128 %[a7], %[a] offset:112\n\t" "ds_read_b128 %[a8], %[a] offset:128\n\t" MASKOP " %[k0], %[k0], %[m]\n\tv_mov_b32 %[f0], %[fv]\n\t" 
#define WV_SLOT_OUTS9 , [a0] "+v"(p[0].a[0]), [a1] "+v"(p[0].a[1]), [a2] "+v"(p[0].a[2]), [a3] "+v"(p[0].a[3]), [a4] "+v"(p[0].a[4]), [a5] "+v"(p[0].a[5]), [a6] "+v"(p[0].a[6]), [a7] "+v"(p[0].a[7]), [a8] "+v"(p[0].a[8]), [k0] "+s"(mask[0]), [f0] "+v"(fl[0])
#define WV_SLOT_BODY10(MASKOP) "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "ds_read_b128 %[a2], %[a] offset:32\n\t" "ds_read_b128 %[a3], %[a] offset:48\n\t" "ds_read_b128 %[a4], %[a] offset:64\n\t" "ds_read_b128 %[a5], %[a] offset:80\n\t" "ds_read_b128 %[a6], %[a] offset:96\n\t" "ds_read_b128 %[a7], %[a] offset:112\n\t" "ds_read_b128 %[a8], %[a] offset:128\n\t" "ds_read_b128 %[a9], %[a] offset:144\n\t" MASKOP " %[k0], %[k0], %[m]\n\tv_mov_b32 %[f0], %[fv]\n\t" 
#define WV_SLOT_OUTS10 , [a0] "+v"(p[0].a[0]), [a1] "+v"(p[0].a[1]), [a2] "+v"(p[0].a[2]), [a3] "+v"(p[0].a[3]), [a4] "+v"(p[0].a[4]), [a5] "+v"(p[0].a[5]), [a6] "+v"(p[0].a[6]), [a7] "+v"(p[0].a[7]), [a8] "+v"(p[0].a[8]), [a9] "+v"(p[0].a[9]), [k0] "+s"(mask[0]), [f0] "+v"(fl[0])
#elif WV_L == 2
#define WV_SLOT_BODY9(MASKOP) "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "ds_read_b128 %[a2], %[a] offset:32\n\t" "ds_read_b128 %[a3], %[a] offset:48\n\t" "ds_read_b128 %[a4], %[a] offset:64\n\t" "ds_read_b128 %[a5], %[a] offset:80\n\t" "ds_read_b128 %[a6], %[a] offset:96\n\t" "ds_read_b128 %[a7], %[a] offset:112\n\t" "ds_read_b128 %[a8], %[a] offset:128\n\t" MASKOP " %[k0], %[k0], %[m]\n\tv_mov_b32 %[f0], %[fv]\n\t" "s_branch 9f\n1:\n\t" "ds_read_b128 %[b0], %[a]\n\t" "ds_read_b128 %[b1], %[a] offset:16\n\t" "ds_read_b128 %[b2], %[a] offset:32\n\t" "ds_read_b128 %[b3], %[a] offset:48\n\t" "ds_read_b128 %[b4], %[a] offset:64\n\t" "ds_read_b128 %[b5], %[a] offset:80\n\t" "ds_read_b128 %[b6], %[a] offset:96\n\t" "ds_read_b128 %[b7], %[a] offset:112\n\t" "ds_read_b128 %[b8], %[a] offset:128\n\t" MASKOP " %[k1], %[k1], %[m]\n\tv_mov_b32 %[f1], %[fv]\n\t" "9:\n\t" 
#define WV_SLOT_OUTS9 , [a0] "+v"(p[0].a[0]), [a1] "+v"(p[0].a[1]), [a2] "+v"(p[0].a[2]), [a3] "+v"(p[0].a[3]), [a4] "+v"(p[0].a[4]), [a5] "+v"(p[0].a[5]), [a6] "+v"(p[0].a[6]), [a7] "+v"(p[0].a[7]), [a8] "+v"(p[0].a[8]), [k0] "+s"(mask[0]), [f0] "+v"(fl[0]), [b0] "+v"(p[1].a[0]), [b1] "+v"(p[1].a[1]), [b2] "+v"(p[1].a[2]), [b3] "+v"(p[1].a[3]), [b4] "+v"(p[1].a[4]), [b5] "+v"(p[1].a[5]), [b6] "+v"(p[1].a[6]), [b7] "+v"(p[1].a[7]), [b8] "+v"(p[1].a[8]), [k1] "+s"(mask[1]), [f1] "+v"(fl[1])
#define WV_SLOT_BODY10(MASKOP) "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "ds_read_b128 %[a2], %[a] offset:32\n\t" "ds_read_b128 %[a3], %[a] offset:48\n\t" "ds_read_b128 %[a4], %[a] offset:64\n\t" "ds_read_b128 %[a5], %[a] offset:80\n\t" "ds_read_b128 %[a6], %[a] offset:96\n\t" "ds_read_b128 %[a7], %[a] offset:112\n\t" "ds_read_b128 %[a8], %[a] offset:128\n\t" "ds_read_b128 %[a9], %[a] offset:144\n\t" MASKOP " %[k0], %[k0], %[m]\n\tv_mov_b32 %[f0], %[fv]\n\t" "s_branch 9f\n1:\n\t" "ds_read_b128 %[b0], %[a]\n\t" "ds_read_b128 %[b1], %[a] offset:16\n\t" "ds_read_b128 %[b2], %[a] offset:32\n\t" "ds_read_b128 %[b3], %[a] offset:48\n\t" "ds_read_b128 %[b4], %[a] offset:64\n\t" "ds_read_b128 %[b5], %[a] offset:80\n\t" "ds_read_b128 %[b6], %[a] offset:96\n\t" "ds_read_b128 %[b7], %[a] offset:112\n\t" "ds_read_b128 %[b8], %[a] offset:128\n\t" "ds_read_b128 %[b9], %[a] offset:144\n\t" MASKOP " %[k1], %[k1], %[m]\n\tv_mov_b32 %[f1], %[fv]\n\t" "9:\n\t" 
#define WV_SLOT_OUTS10 , [a0] "+v"(p[0].a[0]), [a1] "+v"(p[0].a[1]), [a2] "+v"(p[0].a[2]), [a3] "+v"(p[0].a[3]), [a4] "+v"(p[0].a[4]), [a5] "+v"(p[0].a[5]), [a6] "+v"(p[0].a[6]), [a7] "+v"(p[0].a[7]), [a8] "+v"(p[0].a[8]), [a9] "+v"(p[0].a[9]), [k0] "+s"(mask[0]), [f0] "+v"(fl[0]), [b0] "+v"(p[1].a[0]), [b1] "+v"(p[1].a[1]), [b2] "+v"(p[1].a[2]), [b3] "+v"(p[1].a[3]), [b4] "+v"(p[1].a[4]), [b5] "+v"(p[1].a[5]), [b6] "+v"(p[1].a[6]), [b7] "+v"(p[1].a[7]), [b8] "+v"(p[1].a[8]), [b9] "+v"(p[1].a[9]), [k1] "+s"(mask[1]), [f1] "+v"(fl[1])
#elif WV_L == 3
#define WV_SLOT_BODY9(MASKOP) "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "ds_read_b128 %[a2], %[a] offset:32\n\t" "ds_read_b128 %[a3], %[a] offset:48\n\t" "ds_read_b128 %[a4], %[a] offset:64\n\t" "ds_read_b128 %[a5], %[a] offset:80\n\t" "ds_read_b128 %[a6], %[a] offset:96\n\t" "ds_read_b128 %[a7], %[a] offset:112\n\t" "ds_read_b128 %[a8], %[a] offset:128\n\t" MASKOP " %[k0], %[k0], %[m]\n\tv_mov_b32 %[f0], %[fv]\n\t" "s_branch 9f\n1:\n\t" "s_cmp_lg_u32 %[sel], 1\n\ts_cbranch_scc1 2f\n\t" "ds_read_b128 %[b0], %[a]\n\t" "ds_read_b128 %[b1], %[a] offset:16\n\t" "ds_read_b128 %[b2], %[a] offset:32\n\t" "ds_read_b128 %[b3], %[a] offset:48\n\t" "ds_read_b128 %[b4], %[a] offset:64\n\t" "ds_read_b128 %[b5], %[a] offset:80\n\t" "ds_read_b128 %[b6], %[a] offset:96\n\t" "ds_read_b128 %[b7], %[a] offset:112\n\t" "ds_read_b128 %[b8], %[a] offset:128\n\t" MASKOP " %[k1], %[k1], %[m]\n\tv_mov_b32 %[f1], %[fv]\n\t" "s_branch 9f\n2:\n\t" "ds_read_b128 %[c0], %[a]\n\t" "ds_read_b128 %[c1], %[a] offset:16\n\t" "ds_read_b128 %[c2], %[a] offset:32\n\t" "ds_read_b128 %[c3], %[a] offset:48\n\t" "ds_read_b128 %[c4], %[a] offset:64\n\t" "ds_read_b128 %[c5], %[a] offset:80\n\t" "ds_read_b128 %[c6], %[a] offset:96\n\t" "ds_read_b128 %[c7], %[a] offset:112\n\t" "ds_read_b128 %[c8], %[a] offset:128\n\t" MASKOP " %[k2], %[k2], %[m]\n\tv_mov_b32 %[f2], %[fv]\n\t" "9:\n\t" 
#define WV_SLOT_OUTS9 , [a0] "+v"(p[0].a[0]), [a1] "+v"(p[0].a[1]), [a2] "+v"(p[0].a[2]), [a3] "+v"(p[0].a[3]), [a4] "+v"(p[0].a[4]), [a5] "+v"(p[0].a[5]), [a6] "+v"(p[0].a[6]), [a7] "+v"(p[0].a[7]), [a8] "+v"(p[0].a[8]), [k0] "+s"(mask[0]), [f0] "+v"(fl[0]), [b0] "+v"(p[1].a[0]), [b1] "+v"(p[1].a[1]), [b2] "+v"(p[1].a[2]), [b3] "+v"(p[1].a[3]), [b4] "+v"(p[1].a[4]), [b5] "+v"(p[1].a[5]), [b6] "+v"(p[1].a[6]), [b7] "+v"(p[1].a[7]), [b8] "+v"(p[1].a[8]), [k1] "+s"(mask[1]), [f1] "+v"(fl[1]), [c0] "+v"(p[2].a[0]), [c1] "+v"(p[2].a[1]), [c2] "+v"(p[2].a[2]), [c3] "+v"(p[2].a[3]), [c4] "+v"(p[2].a[4]), [c5] "+v"(p[2].a[5]), [c6] "+v"(p[2].a[6]), [c7] "+v"(p[2].a[7]), [c8] "+v"(p[2].a[8]), [k2] "+s"(mask[2]), [f2] "+v"(fl[2])
#define WV_SLOT_BODY10(MASKOP) "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "ds_read_b128 %[a2], %[a] offset:32\n\t" "ds_read_b128 %[a3], %[a] offset:48\n\t" "ds_read_b128 %[a4], %[a] offset:64\n\t" "ds_read_b128 %[a5], %[a] offset:80\n\t" "ds_read_b128 %[a6], %[a] offset:96\n\t" "ds_read_b128 %[a7], %[a] offset:112\n\t" "ds_read_b128 %[a8], %[a] offset:128\n\t" "ds_read_b128 %[a9], %[a] offset:144\n\t" MASKOP " %[k0], %[k0], %[m]\n\tv_mov_b32 %[f0], %[fv]\n\t" "s_branch 9f\n1:\n\t" "s_cmp_lg_u32 %[sel], 1\n\ts_cbranch_scc1 2f\n\t" "ds_read_b128 %[b0], %[a]\n\t" "ds_read_b128 %[b1], %[a] offset:16\n\t" "ds_read_b128 %[b2], %[a] offset:32\n\t" "ds_read_b128 %[b3], %[a] offset:48\n\t" "ds_read_b128 %[b4], %[a] offset:64\n\t" "ds_read_b128 %[b5], %[a] offset:80\n\t" "ds_read_b128 %[b6], %[a] offset:96\n\t" "ds_read_b128 %[b7], %[a] offset:112\n\t" "ds_read_b128 %[b8], %[a] offset:128\n\t" "ds_read_b128 %[b9], %[a] offset:144\n\t" MASKOP " %[k1], %[k1], %[m]\n\tv_mov_b32 %[f1], %[fv]\n\t" "s_branch 9f\n2:\n\t" "ds_read_b128 %[c0], %[a]\n\t" "ds_read_b128 %[c1], %[a] offset:16\n\t" "ds_read_b128 %[c2], %[a] offset:32\n\t" "ds_read_b128 %[c3], %[a] offset:48\n\t" "ds_read_b128 %[c4], %[a] offset:64\n\t" "ds_read_b128 %[c5], %[a] offset:80\n\t" "ds_read_b128 %[c6], %[a] offset:96\n\t" "ds_read_b128 %[c7], %[a] offset:112\n\t" "ds_read_b128 %[c8], %[a] offset:128\n\t" "ds_read_b128 %[c9], %[a] offset:144\n\t" MASKOP " %[k2], %[k2], %[m]\n\tv_mov_b32 %[f2], %[fv]\n\t" "9:\n\t" 
#define WV_SLOT_OUTS10 , [a0] "+v"(p[0].a[0]), [a1] "+v"(p[0].a[1]), [a2] "+v"(p[0].a[2]), [a3] "+v"(p[0].a[3]), [a4] "+v"(p[0].a[4]), [a5] "+v"(p[0].a[5]), [a6] "+v"(p[0].a[6]), [a7] "+v"(p[0].a[7]), [a8] "+v"(p[0].a[8]), [a9] "+v"(p[0].a[9]), [k0] "+s"(mask[0]), [f0] "+v"(fl[0]), [b0] "+v"(p[1].a[0]), [b1] "+v"(p[1].a[1]), [b2] "+v"(p[1].a[2]), [b3] "+v"(p[1].a[3]), [b4] "+v"(p[1].a[4]), [b5] "+v"(p[1].a[5]), [b6] "+v"(p[1].a[6]), [b7] "+v"(p[1].a[7]), [b8] "+v"(p[1].a[8]), [b9] "+v"(p[1].a[9]), [k1] "+s"(mask[1]), [f1] "+v"(fl[1]), [c0] "+v"(p[2].a[0]), [c1] "+v"(p[2].a[1]), [c2] "+v"(p[2].a[2]), [c3] "+v"(p[2].a[3]), [c4] "+v"(p[2].a[4]), [c5] "+v"(p[2].a[5]), [c6] "+v"(p[2].a[6]), [c7] "+v"(p[2].a[7]), [c8] "+v"(p[2].a[8]), [c9] "+v"(p[2].a[9]), [k2] "+s"(mask[2]), [f2] "+v"(fl[2])
#elif WV_L == 4
#define WV_SLOT_BODY9(MASKOP) "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "ds_read_b128 %[a2], %[a] offset:32\n\t" "ds_read_b128 %[a3], %[a] offset:48\n\t" "ds_read_b128 %[a4], %[a] offset:64\n\t" "ds_read_b128 %[a5], %[a] offset:80\n\t" "ds_read_b128 %[a6], %[a] offset:96\n\t" "ds_read_b128 %[a7], %[a] offset:112\n\t" "ds_read_b128 %[a8], %[a] offset:128\n\t" MASKOP " %[k0], %[k0], %[m]\n\tv_mov_b32 %[f0], %[fv]\n\t" "s_branch 9f\n1:\n\t" "s_cmp_lg_u32 %[sel], 1\n\ts_cbranch_scc1 2f\n\t" "ds_read_b128 %[b0], %[a]\n\t" "ds_read_b128 %[b1], %[a] offset:16\n\t" "ds_read_b128 %[b2], %[a] offset:32\n\t" "ds_read_b128 %[b3], %[a] offset:48\n\t" "ds_read_b128 %[b4], %[a] offset:64\n\t" "ds_read_b128 %[b5], %[a] offset:80\n\t" "ds_read_b128 %[b6], %[a] offset:96\n\t" "ds_read_b128 %[b7], %[a] offset:112\n\t" "ds_read_b128 %[b8], %[a] offset:128\n\t" MASKOP " %[k1], %[k1], %[m]\n\tv_mov_b32 %[f1], %[fv]\n\t" "s_branch 9f\n2:\n\t" "s_cmp_lg_u32 %[sel], 2\n\ts_cbranch_scc1 3f\n\t" "ds_read_b128 %[c0], %[a]\n\t" "ds_read_b128 %[c1], %[a] offset:16\n\t" "ds_read_b128 %[c2], %[a] offset:32\n\t" "ds_read_b128 %[c3], %[a] offset:48\n\t" "ds_read_b128 %[c4], %[a] offset:64\n\t" "ds_read_b128 %[c5], %[a] offset:80\n\t" "ds_read_b128 %[c6], %[a] offset:96\n\t" "ds_read_b128 %[c7], %[a] offset:112\n\t" "ds_read_b128 %[c8], %[a] offset:128\n\t" MASKOP " %[k2], %[k2], %[m]\n\tv_mov_b32 %[f2], %[fv]\n\t" "s_branch 9f\n3:\n\t" "ds_read_b128 %[d0], %[a]\n\t" "ds_read_b128 %[d1], %[a] offset:16\n\t" "ds_read_b128 %[d2], %[a] offset:32\n\t" "ds_read_b128 %[d3], %[a] offset:48\n\t" "ds_read_b128 %[d4], %[a] offset:64\n\t" "ds_read_b128 %[d5], %[a] offset:80\n\t" "ds_read_b128 %[d6], %[a] offset:96\n\t" "ds_read_b128 %[d7], %[a] offset:112\n\t" "ds_read_b128 %[d8], %[a] offset:128\n\t" MASKOP " %[k3], %[k3], %[m]\n\tv_mov_b32 %[f3], %[fv]\n\t" "9:\n\t" 
#define WV_SLOT_OUTS9 , [a0] "+v"(p[0].a[0]), [a1] "+v"(p[0].a[1]), [a2] "+v"(p[0].a[2]), [a3] "+v"(p[0].a[3]), [a4] "+v"(p[0].a[4]), [a5] "+v"(p[0].a[5]), [a6] "+v"(p[0].a[6]), [a7] "+v"(p[0].a[7]), [a8] "+v"(p[0].a[8]), [k0] "+s"(mask[0]), [f0] "+v"(fl[0]), [b0] "+v"(p[1].a[0]), [b1] "+v"(p[1].a[1]), [b2] "+v"(p[1].a[2]), [b3] "+v"(p[1].a[3]), [b4] "+v"(p[1].a[4]), [b5] "+v"(p[1].a[5]), [b6] "+v"(p[1].a[6]), [b7] "+v"(p[1].a[7]), [b8] "+v"(p[1].a[8]), [k1] "+s"(mask[1]), [f1] "+v"(fl[1]), [c0] "+v"(p[2].a[0]), [c1] "+v"(p[2].a[1]), [c2] "+v"(p[2].a[2]), [c3] "+v"(p[2].a[3]), [c4] "+v"(p[2].a[4]), [c5] "+v"(p[2].a[5]), [c6] "+v"(p[2].a[6]), [c7] "+v"(p[2].a[7]), [c8] "+v"(p[2].a[8]), [k2] "+s"(mask[2]), [f2] "+v"(fl[2]), [d0] "+v"(p[3].a[0]), [d1] "+v"(p[3].a[1]), [d2] "+v"(p[3].a[2]), [d3] "+v"(p[3].a[3]), [d4] "+v"(p[3].a[4]), [d5] "+v"(p[3].a[5]), [d6] "+v"(p[3].a[6]), [d7] "+v"(p[3].a[7]), [d8] "+v"(p[3].a[8]), [k3] "+s"(mask[3]), [f3] "+v"(fl[3])
#define WV_SLOT_BODY10(MASKOP) "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "ds_read_b128 %[a2], %[a] offset:32\n\t" "ds_read_b128 %[a3], %[a] offset:48\n\t" "ds_read_b128 %[a4], %[a] offset:64\n\t" "ds_read_b128 %[a5], %[a] offset:80\n\t" "ds_read_b128 %[a6], %[a] offset:96\n\t" "ds_read_b128 %[a7], %[a] offset:112\n\t" "ds_read_b128 %[a8], %[a] offset:128\n\t" "ds_read_b128 %[a9], %[a] offset:144\n\t" MASKOP " %[k0], %[k0], %[m]\n\tv_mov_b32 %[f0], %[fv]\n\t" "s_branch 9f\n1:\n\t" "s_cmp_lg_u32 %[sel], 1\n\ts_cbranch_scc1 2f\n\t" "ds_read_b128 %[b0], %[a]\n\t" "ds_read_b128 %[b1], %[a] offset:16\n\t" "ds_read_b128 %[b2], %[a] offset:32\n\t" "ds_read_b128 %[b3], %[a] offset:48\n\t" "ds_read_b128 %[b4], %[a] offset:64\n\t" "ds_read_b128 %[b5], %[a] offset:80\n\t" "ds_read_b128 %[b6], %[a] offset:96\n\t" "ds_read_b128 %[b7], %[a] offset:112\n\t" "ds_read_b128 %[b8], %[a] offset:128\n\t" "ds_read_b128 %[b9], %[a] offset:144\n\t" MASKOP " %[k1], %[k1], %[m]\n\tv_mov_b32 %[f1], %[fv]\n\t" "s_branch 9f\n2:\n\t" "s_cmp_lg_u32 %[sel], 2\n\ts_cbranch_scc1 3f\n\t" "ds_read_b128 %[c0], %[a]\n\t" "ds_read_b128 %[c1], %[a] offset:16\n\t" "ds_read_b128 %[c2], %[a] offset:32\n\t" "ds_read_b128 %[c3], %[a] offset:48\n\t" "ds_read_b128 %[c4], %[a] offset:64\n\t" "ds_read_b128 %[c5], %[a] offset:80\n\t" "ds_read_b128 %[c6], %[a] offset:96\n\t" "ds_read_b128 %[c7], %[a] offset:112\n\t" "ds_read_b128 %[c8], %[a] offset:128\n\t" "ds_read_b128 %[c9], %[a] offset:144\n\t" MASKOP " %[k2], %[k2], %[m]\n\tv_mov_b32 %[f2], %[fv]\n\t" "s_branch 9f\n3:\n\t" "ds_read_b128 %[d0], %[a]\n\t" "ds_read_b128 %[d1], %[a] offset:16\n\t" "ds_read_b128 %[d2], %[a] offset:32\n\t" "ds_read_b128 %[d3], %[a] offset:48\n\t" "ds_read_b128 %[d4], %[a] offset:64\n\t" "ds_read_b128 %[d5], %[a] offset:80\n\t" "ds_read_b128 %[d6], %[a] offset:96\n\t" "ds_read_b128 %[d7], %[a] offset:112\n\t" "ds_read_b128 %[d8], %[a] offset:128\n\t" "ds_read_b128 %[d9], %[a] offset:144\n\t" MASKOP " %[k3], %[k3], %[m]\n\tv_mov_b32 %[f3], %[fv]\n\t" "9:\n\t" 
#define WV_SLOT_OUTS10 , [a0] "+v"(p[0].a[0]), [a1] "+v"(p[0].a[1]), [a2] "+v"(p[0].a[2]), [a3] "+v"(p[0].a[3]), [a4] "+v"(p[0].a[4]), [a5] "+v"(p[0].a[5]), [a6] "+v"(p[0].a[6]), [a7] "+v"(p[0].a[7]), [a8] "+v"(p[0].a[8]), [a9] "+v"(p[0].a[9]), [k0] "+s"(mask[0]), [f0] "+v"(fl[0]), [b0] "+v"(p[1].a[0]), [b1] "+v"(p[1].a[1]), [b2] "+v"(p[1].a[2]), [b3] "+v"(p[1].a[3]), [b4] "+v"(p[1].a[4]), [b5] "+v"(p[1].a[5]), [b6] "+v"(p[1].a[6]), [b7] "+v"(p[1].a[7]), [b8] "+v"(p[1].a[8]), [b9] "+v"(p[1].a[9]), [k1] "+s"(mask[1]), [f1] "+v"(fl[1]), [c0] "+v"(p[2].a[0]), [c1] "+v"(p[2].a[1]), [c2] "+v"(p[2].a[2]), [c3] "+v"(p[2].a[3]), [c4] "+v"(p[2].a[4]), [c5] "+v"(p[2].a[5]), [c6] "+v"(p[2].a[6]), [c7] "+v"(p[2].a[7]), [c8] "+v"(p[2].a[8]), [c9] "+v"(p[2].a[9]), [k2] "+s"(mask[2]), [f2] "+v"(fl[2]), [d0] "+v"(p[3].a[0]), [d1] "+v"(p[3].a[1]), [d2] "+v"(p[3].a[2]), [d3] "+v"(p[3].a[3]), [d4] "+v"(p[3].a[4]), [d5] "+v"(p[3].a[5]), [d6] "+v"(p[3].a[6]), [d7] "+v"(p[3].a[7]), [d8] "+v"(p[3].a[8]), [d9] "+v"(p[3].a[9]), [k3] "+s"(mask[3]), [f3] "+v"(fl[3])
#endif
#endif /* WV_HDP */
/* (the row's tenth pair -- the switch sum and the raw gap-X emission -- is only loaded by the builds that use it) */
template <bool SW> __device__ __forceinline__ void slot_enter(Prm (&p)[WV_L], unsigned long long (&mask)[WV_L],
                                                             unsigned (&fl)[WV_L], unsigned rowAddr,
                                                             unsigned long long laneMask, int sel) {
    if (SW) { WV_SLOT_ASM("s_or_b64", 1, WV_SLOT_BODY10, WV_SLOT_OUTS10) }
    else { WV_SLOT_ASM("s_or_b64", 1, WV_SLOT_BODY9, WV_SLOT_OUTS9) }
}
template <bool SW> __device__ __forceinline__ void slot_leave(Prm (&p)[WV_L], unsigned long long (&mask)[WV_L],
                                                             unsigned (&fl)[WV_L], unsigned rowAddr,
                                                             unsigned long long laneMask, int sel) {
    if (SW) { WV_SLOT_ASM("s_andn2_b64", 0, WV_SLOT_BODY10, WV_SLOT_OUTS10) }
    else { WV_SLOT_ASM("s_andn2_b64", 0, WV_SLOT_BODY9, WV_SLOT_OUTS9) }
}


/* the band: first and last matrix column (k-mer index) of every anti-diagonal, one int2 per diagonal in HBM
 * (built by the host from band_construct's output) */
__device__ __forceinline__ void band_load(const int2 *__restrict__ tab, int d, int &xmin, int &xmax) {
    const int2 v = tab[d > 0 ? d : 0];
    xmin = uni(v.x);
    xmax = uni(v.y);
}

/* The edge steps of diagonals lo..hi into the LDS bit strings: bit d of stepMin is xmin(d) - xmin(d-1),
 * bit d of stepMax is xmax(d) - xmax(d-1) (both 0 or 1: the host checked).  Each lane takes 16
 * consecutive diagonals per round (17 table entries, all loads in flight together). */
__device__ void stage_band_steps(unsigned (&bits)[2][WV_BITWORDS], const int2 *__restrict__ tab, int D, int lo, int hi) {
    const int lane = threadIdx.x & 63;
    for (int w = (lo >> 5) + lane; w <= (hi >> 5); w += 64) {
        bits[0][w & (WV_BITWORDS - 1)] = 0u;
        bits[1][w & (WV_BITWORDS - 1)] = 0u;
    }
    for (int base = lo & ~31; base <= hi; base += 1024) {
        const int d0 = base + lane * 16;
        int2 e[17];
#pragma unroll
        for (int k = 0; k < 17; k++) {
            int d = d0 - 1 + k;
            d = d < 0 ? 0 : d > D ? D : d;
            e[k] = tab[d];
        }
        unsigned mn = 0u, mx = 0u;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const bool in = d0 + k >= lo && d0 + k <= hi && d0 + k >= 1;
            mn |= (in && e[k + 1].x != e[k].x ? 1u : 0u) << k;
            mx |= (in && e[k + 1].y != e[k].y ? 1u : 0u) << k;
        }
        if (d0 <= hi) {
            const int w = (d0 >> 5) & (WV_BITWORDS - 1), sh = (lane & 1) * 16;
            if (mn) atomicOr(&bits[0][w], mn << sh);
            if (mx) atomicOr(&bits[1][w], mx << sh);
        }
    }
}

/* The next traceback point (:917-921) from the band alone: the first diagonal above dAfter that is at
 * least dMin and narrow enough, or the last diagonal D.  512 diagonals per round. */
__device__ int next_traceback_point(const int2 *__restrict__ tab, int D, int dAfter, long long dMin, long long widthLimit) {
    const long long b0 = dAfter + 1 > dMin ? dAfter + 1 : dMin;
    if (b0 >= D) return D;
    const int lane = threadIdx.x & 63;
    for (int base = (int) b0;; base += 512) {
        int first = 8;
#pragma unroll
        for (int k = 7; k >= 0; k--) {
            const int d = base + lane * 8 + k;
            if (d >= D) first = k; /* the last diagonal is a traceback point whatever its width */
            else {
                const int2 v = tab[d];
                if ((long long) (v.y - v.x + 1) <= widthLimit) first = k;
            }
        }
        const unsigned long long m = __ballot(first < 8);
        if (m != 0ull) {
            const int l0 = __ffsll((long long) m) - 1;
            const int d = base + l0 * 8 + __builtin_amdgcn_readlane(first, l0);
            return d < D ? d : D;
        }
    }
}

struct FwdShared {
    double coef[64];
    double ev[(2 * WV_P + WV_L) * 2];       /* events (mean, noise) by index mod P, mirrored */
#ifdef WV_VANILLA
    double ev2[(2 * WV_P + WV_L) * 2];      /* ... and (1 / noise, log noise), same indexing */
#endif
    double rows[(WV_ROWN + 1) * WV_ROW];    /* k-mer rows by column mod WV_ROWN; row WV_ROWN = "not a k-mer" */
    unsigned bits[2][WV_BITWORDS];
};

/*
 * Forward sweep of one alignment from its saved diagonal up to (and including) the next traceback
 * point; describes the window for the backward kernel.
 */
template <bool SW> __device__ void forward_window(const DevItem &it, const DevParams &P, const int2 *__restrict__ bandTab,
                               const double *__restrict__ track, const double *__restrict__ events,
                               const double *__restrict__ model, double *ring, int ringD, WvState *state,
                               int window, FwdShared &sh) {
    constexpr int L = WV_L;
    const int lane = threadIdx.x & 63;
    const int lX = (int) it.lX, lY = (int) it.lY, D = lX + lY;
    const double *__restrict__ ev = events + 3 * it.yOff;
    const unsigned cf = lds_addr(sh.coef);
    const unsigned rowsAddr = lds_addr(sh.rows), parkAddr = rowsAddr + WV_ROWN * WV_ROW * 8;
    const int ringMask = ringD - 1;
    double T[9];
#pragma unroll
    for (int i = 0; i < 9; i++) T[i] = model[i];
#ifdef WV_VANILLA
    /* the position-independent part of the machine: a_ym, a_yy and the end vector (cpecan_hip.hip: derive_vanilla) */
    const double lYM = uni64_d(model[CP_VHDR_LOG_YM]), lYY = uni64_d(model[CP_VHDR_LOG_YY]);
    const double endM = uni64_d(model[CP_VHDR_END_M]), endX = uni64_d(model[CP_VHDR_END_X]),
                 endY = uni64_d(model[CP_VHDR_END_Y]);
#endif
#ifdef WV_HDP
    /* the NanoporeHDP as densities need it (dir_proc_density impl/hdp.c:2577-2601 -> grid_spline_interp
     * impl/hdp_math_utils.c:471-495): an evenly spaced sampling grid, values and spline slopes per table row */
    const DevHdpModel *hm = (const DevHdpModel *) model;
    const double *__restrict__ hGrid = hm->grid, *__restrict__ hY = hm->y, *__restrict__ hS = hm->slope;
    const int hN = uni(hm->gridLength) - 1;
    const double gX0 = uni64_d(hGrid[0]), gXn = uni64_d(hGrid[hN]), gDx = uni64_d(hGrid[1] - hGrid[0]);
#endif
    /* the feed block: events of a block are staged one block ahead of their first use, so the ring of P
     * events must hold the band's events and a block's worth more */
    int feed = WV_P - it.maxWidth;
    feed = feed > WV_FEED_MAX ? WV_FEED_MAX : feed & ~1;

    const int d0 = uni(ld_agent(&state->d));
    const int tracedBackTo = uni(ld_agent(&state->tracedBackTo));
    long long cells = uni64(ld_agent(&state->cells));

    /* the "not a k-mer" row: every emission -inf */
    if (lane < WV_ROW) {
#if defined(WV_VANILLA)
        /* (level K -inf; the noise term kept finite: mean 1, 1/mean 1, lambda 1, c1 0; log transitions -inf) */
        sh.rows[WV_ROWN * WV_ROW + lane] = lane >= 16 || lane == 3 || lane == 11 ? CP_NEG_INF
                                           : (lane & 7) == 4 || (lane & 7) == 5 || (lane & 7) == 6 ? 1.0 : 0.0;
#elif defined(WV_HDP)
        /* (table row offset -1: the density of a parked slot is -inf; gap-X sums -inf) */
        sh.rows[WV_ROWN * WV_ROW + lane] = lane >= 16 ? CP_NEG_INF : lane == 0 ? -1.0 : 0.0;
#else
        const bool inf = lane == CP_K1 || lane == CP_K2 || lane == CP_YK1 || lane == CP_YK2 || lane >= 16;
        sh.rows[WV_ROWN * WV_ROW + lane] = inf ? CP_NEG_INF : 0.0;
#endif
    }
#ifdef WV_VANILLA
    /* an event that does not exist scores as (mean 0, noise 1): finite, and it only ever meets -inf cells */
    for (int i = lane; i < (2 * WV_P + WV_L) * 2; i += 64) {
        sh.ev[i] = (i & 1) ? 1.0 : 0.0;
        sh.ev2[i] = (i & 1) ? 0.0 : 1.0;
    }
    const unsigned ev2Delta = lds_addr(sh.ev2) - lds_addr(sh.ev);
#else
    for (int i = lane; i < (2 * WV_P + WV_L) * 2; i += 64) sh.ev[i] = 0.0;
#endif

    /* ---- per-slot state: layer j of this lane is slot lane * L + j ---- */
    Prm prm[L];
    double Am[L], Ax[L], Ay[L]; /* forward cells of the last diagonal done      */
    double Bm[L], Bx[L], By[L]; /* ... and of the one before                    */
    double RAm, RAx, RAy, RBm, RBx, RBy; /* layer L-1 of the lane below, same two diagonals */
    unsigned long long mask[L];  /* lanes whose slot holds a k-mer of the band   */
    unsigned fl[L];              /* the same, per lane: 1 or 0                    */
    int xmin, xmax;              /* band of the last diagonal done               */
    int inL, inJ, outL, outJ;    /* slots of k-mers xmax + 1 (next to enter) and xmin (next to leave) */

    if (d0 == 0) {
        /* diagonal 0: the single cell (0,0) holds the start vector (:897-898, stateMachine.c:1168-1177) */
#pragma unroll
        for (int j = 0; j < L; j++) {
            load_row_all(prm[j], parkAddr);
            Am[j] = Ax[j] = Ay[j] = Bm[j] = Bx[j] = By[j] = CP_NEG_INF;
            mask[j] = 0ull;
            fl[j] = 0u;
        }
        if (lane == 0) {
            Am[0] = it.raggedL ? CP_NEG_INF : 0.0;
            Ax[0] = it.raggedL ? 0.0 : CP_NEG_INF;
            Ay[0] = Ax[0];
        }
        mask[0] = 1ull;
        if (lane == 0) {
            fl[0] = 1u;
            ring[WV_OFF_FM(0)] = Am[0]; ring[WV_OFF_PM(0)] = 0.0; ring[WV_ROW_PY(0, 0)] = 0.0;
            ring[WV_OFF_FX(0)] = Ax[0]; ring[WV_OFF_FY(0)] = Ay[0];
        }
#if defined(WV_HDP) || defined(WV_VANILLA)
        /* matrix column 0 scores the FIRST k-mer (pair) under this machine (sequence_getKmer3 / sequence_getKmer2,
         * :320-331: index -1 reads element 0), not a sentinel: its slot starts with that row */
        if (lane == 0) {
            const d2 *src = (const d2 *) track;
#pragma unroll
            for (int k = 0; k < WV_ROW / 2; k++) prm[0].a[k] = src[k];
        }
#endif
        /* the dummy row the sweep back reads for lanes without a cell: -inf everywhere */
        for (int i = lane; i < WV_ROW_DOUBLES; i += 64) ring[(long long) ringD * WV_ROW_DOUBLES + i] = CP_NEG_INF;
        RBm = RBx = RBy = CP_NEG_INF;
        cells += 1;
        xmin = xmax = 0;
        inL = 1 / L; inJ = 1 % L;
        outL = 0; outJ = 0;
        /* column 0 scores the "not a k-mer" sentinel (sequence_getKmer index -1, :314-318): its row is the parked one */
    } else {
        /* resume at d0: constants of the k-mers in the band, forward cells of d0 and d0-1 */
        int qmin, qmax;
        band_load(bandTab, d0 - 1, qmin, qmax);
        band_load(bandTab, d0, xmin, xmax);
        const double *r1 = ring + (long long) (d0 & ringMask) * WV_ROW_DOUBLES;
        const double *r2 = ring + (long long) ((d0 - 1) & ringMask) * WV_ROW_DOUBLES;
#pragma unroll
        for (int j = 0; j < L; j++) {
            const int s = lane * L + j;
            int xs = s + ((xmin - s + WV_P - 1) / WV_P) * WV_P; /* the k-mer >= xmin that lives in this slot */
            const bool v1 = xs <= xmax;
            int xq = s + ((qmin - s + WV_P - 1) / WV_P) * WV_P;
            const bool v2 = xq <= qmax;
            load_row_all(prm[j], parkAddr);
            if (v1) {
                const d2 *src = (const d2 *) (track + (long long) xs * WV_ROW);
#pragma unroll
                for (int k = 0; k < WV_ROW / 2; k++) prm[j].a[k] = src[k];
            }
            /* ring slots of cells outside the band hold stale data: mask per lane */
            Am[j] = v1 ? r1[j * WV_LAYER_DOUBLES + WV_OFF_FM(lane)] : CP_NEG_INF;
            Ax[j] = v1 ? r1[j * WV_LAYER_DOUBLES + WV_OFF_FX(lane)] : CP_NEG_INF;
            Ay[j] = v1 ? r1[j * WV_LAYER_DOUBLES + WV_OFF_FY(lane)] : CP_NEG_INF;
            Bm[j] = v2 ? r2[j * WV_LAYER_DOUBLES + WV_OFF_FM(lane)] : CP_NEG_INF;
            Bx[j] = v2 ? r2[j * WV_LAYER_DOUBLES + WV_OFF_FX(lane)] : CP_NEG_INF;
            By[j] = v2 ? r2[j * WV_LAYER_DOUBLES + WV_OFF_FY(lane)] : CP_NEG_INF;
            mask[j] = (unsigned long long) uni64((long long) __ballot(v1));
            fl[j] = v1 ? 1u : 0u;
        }
        RBm = ror1(Bm[L - 1]); RBx = ror1(Bx[L - 1]); RBy = ror1(By[L - 1]);
        const int si = (xmax + 1) % WV_P, so = xmin % WV_P;
        inL = uni(si / L); inJ = uni(si % L);
        outL = uni(so / L); outJ = uni(so % L);
    }
    RAm = RAx = RAy = CP_NEG_INF;

    /*
     * Which diagonals need all three states in the ring.  The sweep back reads only the match cell of
     * a diagonal, except where it refreshes totalProbability (every 10th decoded diagonal, counted
     * down from the first one of ITS window: it then reads every state of that diagonal and of the
     * one below), and this sweep resumes from the last two diagonals of a launch.  Where the windows
     * will start is a function of the band alone, so it is known here: this launch ends at topW; its
     * diagonals up to fromW are decoded by window W (first decoded diagonal tpA), the ones above by
     * the next window (tpB).  Everywhere else the two gap states are not stored.
     */
    const long long widthLimit = P.expansion * 2 + 1;
    const int topW = next_traceback_point(bandTab, D, d0, tracedBackTo + P.minDiags, widthLimit);
    const bool endW = topW == D;
    const int fromW = topW - (endW ? 0 : (int) P.tbDiags + 1);
    const int tpA = topW < fromW ? topW : fromW;
    int tpB = tpA;
    bool allFull = false;
    if (!endW) {
        const int topN = next_traceback_point(bandTab, D, topW, fromW + P.minDiags, widthLimit);
        const int fromN = topN - (topN == D ? 0 : (int) P.tbDiags + 1);
        tpB = topN < fromN ? topN : fromN;
        allFull = tpB < topW; /* windows shorter than the traceback margin: keep everything */
    }
    if (P.mode != 0) allFull = true; /* the expectation pass reads every state of every diagonal */
    int rA = ((tpA - (d0 + 1)) % 10 + 10) % 10, rB = endW ? 0x40000000 : ((tpB - (d0 + 1)) % 10 + 10) % 10;
    const int fullFrom = allFull ? -0x40000000 : topW - 1;

    /* the circular bit strings hold 8192 diagonals: longer launches are staged 4096 diagonals (whole words) at a time */
    int bitsHi = topW - d0 > 4096 ? ((d0 + 1 + 4096) & ~31) - 1 : topW;
    stage_band_steps(sh.bits, bandTab, D, d0 + 1, bitsHi);

    int evHi = d0 - xmax - 1;  /* first event not yet staged: the lowest index diagonal d0 + 1 can ask for */
    int evHiMod = ((evHi % WV_P) + WV_P) % WV_P;
    int rowHi = xmax + 1;      /* first k-mer row not yet staged */
    unsigned dmod16 = (unsigned) ((d0 + 1) % WV_P) * 16u; /* 16 * (d mod P) of the diagonal being computed */
    /* byte address of this lane's layer-(L-1) event on a diagonal d: 16 * ((d - L * (lane + 1)) mod P), split
     * into the wave-uniform 16 * (d mod P) and a lane constant; the ring is mirrored, so the sum needs no wrap */
    const unsigned evLane = lds_addr(sh.ev) + 16u * (unsigned) ((WV_P - L * (lane + 1) % WV_P) % WV_P);
    unsigned wMin = 0u, wMax = 0u;

    auto step = [&](const int d, const bool full, double (&cm)[L], double (&cx)[L], double (&cy)[L],
                    double (&qm)[L], double (&qx)[L], double (&qy)[L], double &rlm, double &rlx, double &rly,
                    const double rmm, const double rmx, const double rmy) __attribute__((always_inline)) {
        /* the band of diagonal d: leaving k-mer first (its slot is parked), then the entering one */
        const unsigned bi = (unsigned) d & 31u;
        if (bi == 0u || d == d0 + 1) {
            wMin = (unsigned) uni((int) sh.bits[0][(d >> 5) & (WV_BITWORDS - 1)]);
            wMax = (unsigned) uni((int) sh.bits[1][(d >> 5) & (WV_BITWORDS - 1)]);
        }
        if ((wMin >> bi) & 1u) {
            slot_leave<SW>(prm, mask, fl, parkAddr, 1ull << outL, outJ);
            xmin++;
            if (++outJ == L) { outJ = 0; outL = (outL + 1) & 63; }
        }
        if ((wMax >> bi) & 1u) {
            xmax++;
            const unsigned ra = rowsAddr + (unsigned) (xmax & (WV_ROWN - 1)) * (WV_ROW * 8);
            slot_enter<SW>(prm, mask, fl, ra, 1ull << inL, inJ);
            if (++inJ == L) { inJ = 0; inL = (inL + 1) & 63; }
        }
        cells += xmax - xmin + 1;
        /* layer L-1 of the lane below, last diagonal (its value of the diagonal before is rm*) */
        rlm = ror1(cm[L - 1]); rlx = ror1(cx[L - 1]); rly = ror1(cy[L - 1]);
        const unsigned ea = evLane + dmod16;
        dmod16 = dmod16 + 16u == WV_P * 16u ? 0u : dmod16 + 16u;
        double *rowBase = ring + (long long) (d & ringMask) * WV_ROW_DOUBLES;
        d2 e[L];
#pragma unroll
        for (int j = 0; j < L; j++) e[j] = *lds_d2p_cast(ea + 16u * (unsigned) (L - 1 - j));
        /* cell_calculateForward: to[t] = logAdd(to[t], from[f] + (eP + tP)) (:365-376) in the order of
         * stateMachine3_cellCalculate (stateMachine.c:1314-1333); the gap-X sums eP + tP are the row's (built with
         * the same additions by the track kernel).  The L cells of a lane are independent: their logAdds are
         * taken stage by stage, so that the L table reads of a stage are in flight together. */
        double pm[L], py[L], nm[L], nx[L], ny[L], t1[L];
#if defined(WV_VANILLA)
        /* emissions_signal_getEventMatchProbWithTwoDists (impl/stateMachine.c:499-528) on the match table and on the
         * extra-event table: logGaussPdf of the mean (:333-343) + logInvGaussPdf of the noise (:322-331),
         * ((log(lambda) - log(2 pi) - 3 log(noise)) - lambda a a / noise) / 2 with a = (noise - mu) / mu; both
         * divisions as Markstein-corrected multiplies by host-/device-rounded reciprocals, log(noise) from the host */
        d2 e2[L];
#pragma unroll
        for (int j = 0; j < L; j++) e2[j] = *lds_d2p_cast(ea + ev2Delta + 16u * (unsigned) (L - 1 - j));
#pragma unroll
        for (int j = 0; j < L; j++) {
            const Prm &p = prm[j];
            const double c3 = 3 * e2[j].y;
#pragma unroll
            for (int tb = 0; tb < 2; tb++) {
                const d2 g0 = p.a[4 * tb], g1 = p.a[4 * tb + 1], n0 = p.a[4 * tb + 2], n1 = p.a[4 * tb + 3];
                const double level = lgauss(e[j].x, g0.x, g0.y, g1.x, g1.y);
                const double u = e[j].y - n0.x;
                const double q1 = u * n0.y;
                const double a = __fma_rn(__fma_rn(-q1, n0.x, u), n0.y, q1); /* (noise - mu) / mu */
                const double w = n1.x * a * a;
                const double q2 = w * e2[j].x;
                const double dv = __fma_rn(__fma_rn(-q2, e[j].y, w), e2[j].x, q2); /* lambda a a / noise */
                const double nz = (n1.y - c3 - dv) / 2;
                if (tb == 0) pm[j] = level + nz;
                else py[j] = level + nz;
            }
        }
#elif defined(WV_HDP)
        /* get_nanopore_kmer_density (impl/nanopore_hdp.c:390): the spline of the slot's table row at the cell's event,
         * clamped at zero -- a linear density where a log-probability belongs (quirk Q1), match and gap-Y emission
         * alike.  The event's grid cell and its offset in it were taken when the event was staged. */
        double hYl[L], hYr[L], hSl[L], hSr[L];
#pragma unroll
        for (int j = 0; j < L; j++) {
            const double ro = prm[j].a[0].x, ex = e[j].x;
            const unsigned base = ro < 0.0 ? 0u : (unsigned) ro;
            const unsigned il = ex < 0.0 ? (ex == -2.0 ? (unsigned) hN : 0u) : (unsigned) ex;
            const unsigned ir = ex < 0.0 ? il : il + 1u;
            hYl[j] = hY[base + il]; hYr[j] = hY[base + ir];
            hSl[j] = hS[base + il]; hSr[j] = hS[base + ir];
        }
#pragma unroll
        for (int j = 0; j < L; j++) {
            const double ex = e[j].x, w = e[j].y;
            const double dy = hYr[j] - hYl[j];
            const double ca = hSl[j] * gDx - dy, cb = dy - hSr[j] * gDx;
            const double tl = w, tr = 1.0 - tl;
            const double inside = tr * hYl[j] + tl * hYr[j] + tl * tr * (ca * tr + cb * tl);
            const double edge = ex == -1.0 ? hYl[j] - hSl[j] * w : hYl[j] + hSl[j] * w;
            double r = ex < 0.0 ? edge : inside;
            r = r > 0.0 ? r : 0.0;
            pm[j] = py[j] = prm[j].a[0].x < 0.0 ? CP_NEG_INF : r;
        }
#else
#pragma unroll
        for (int j = 0; j < L; j++) {
            const Prm &p = prm[j];
            pm[j] = lgauss(e[j].x, p.a[0].x, p.a[0].y, p.a[1].x, p.a[1].y)
                    + lgauss(e[j].y, p.a[2].x, p.a[2].y, p.a[3].x, p.a[3].y);
            py[j] = lgauss(e[j].x, p.a[4].x, p.a[4].y, p.a[5].x, p.a[5].y)
                    + lgauss(e[j].y, p.a[6].x, p.a[6].y, p.a[7].x, p.a[7].y);
        }
#endif /* WV_HDP */
        double t2[L], t3[L], t4[L];
        LaddPending<L> pa, pb;
#pragma unroll
        for (int j = 0; j < L; j++) { /* gap X from the lower cell: (x-1, y) on the last diagonal */
            nx[j] = (j ? cm[j ? j - 1 : 0] : rlm) + prm[j].a[8].x;
            t1[j] = (j ? cx[j ? j - 1 : 0] : rlx) + prm[j].a[8].y;
        }
        ladd_issue<L>(pa, nx, t1, cf);
#pragma unroll
        for (int j = 0; j < L; j++) { /* match from the middle cell: (x-1, y-1) on the diagonal before */
#ifdef WV_VANILLA
            nm[j] = (j ? qm[j ? j - 1 : 0] : rmm) + (pm[j] + prm[j].a[9].x);
            t2[j] = (j ? qx[j ? j - 1 : 0] : rmx) + (pm[j] + prm[j].a[9].y);
#else
            nm[j] = (j ? qm[j ? j - 1 : 0] : rmm) + (pm[j] + T[T_MATCH_CONTINUE]);
            t2[j] = (j ? qx[j ? j - 1 : 0] : rmx) + (pm[j] + T[T_MATCH_FROM_GAP_X]);
#endif
        }
        ladd_issue<L>(pb, nm, t2, cf);
        ladd_finish<L>(pa, nx);
#pragma unroll
        for (int j = 0; j < L; j++) { /* gap Y from the upper cell: (x, y-1) on the last diagonal */
#ifdef WV_VANILLA
            ny[j] = cm[j] + (py[j] + prm[j].a[10].x);
            t3[j] = cy[j] + (py[j] + lYY);
#else
            ny[j] = cm[j] + (py[j] + T[T_GAP_OPEN_Y]);
            t3[j] = cy[j] + (py[j] + T[T_GAP_EXTEND_Y]);
#endif
        }
        ladd_issue<L>(pa, ny, t3, cf);
        ladd_finish<L>(pb, nm);
#pragma unroll
#ifdef WV_VANILLA
        for (int j = 0; j < L; j++) t4[j] = (j ? qy[j ? j - 1 : 0] : rmy) + (pm[j] + lYM);
#else
        for (int j = 0; j < L; j++) t4[j] = (j ? qy[j ? j - 1 : 0] : rmy) + (pm[j] + T[T_MATCH_FROM_GAP_Y]);
#endif
        ladd_issue<L>(pb, nm, t4, cf);
        ladd_finish<L>(pa, ny);
        if (SW) {
#pragma unroll
            for (int j = 0; j < L; j++) t1[j] = (j ? cy[j ? j - 1 : 0] : rly) + prm[j].a[9].x;
            ladd_issue<L>(pa, nx, t1, cf);
        }
        ladd_finish<L>(pb, nm);
        if (SW) ladd_finish<L>(pa, nx);
#pragma unroll
        for (int j = 0; j < L; j++) {
            /* only cells of the band go to the ring (plain predicated stores: the compiler spreads them over the
             * arithmetic; exec-masking them by hand in asm blocks at the end of the step cost a third of the sweep) */
            if (fl[j] != 0u) {
                double *q = rowBase + j * WV_LAYER_DOUBLES;
                d2 pr;
                pr.x = nm[j]; pr.y = pm[j];
                *(d2 *) (q + WV_OFF_FM(lane)) = pr;
                rowBase[WV_ROW_PY(j, lane)] = py[j];
                if (full) {
                    d2 gxy;
                    gxy.x = nx[j]; gxy.y = ny[j];
                    *(d2 *) (q + WV_OFF_FX(lane)) = gxy;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < L; j++) { qm[j] = nm[j]; qx[j] = nx[j]; qy[j] = ny[j]; }
    };

    const long long tbFromL = tracedBackTo + P.minDiags;
    const int tbFrom = (int) (tbFromL < 0x7fffffff ? tbFromL : 0x7fffffff);
    const int tbWidth = (int) (widthLimit < 0x7fffffff ? widthLimit : 0x7fffffff);
    auto finish = [&](const int d, const double (&fm)[L], const double (&fx)[L], const double (&fy)[L])
                      __attribute__((always_inline)) {
        /* traceback point (:917-921) reached: hand the window to the backward kernel */
        const bool atEnd = d == D;
        if (!(atEnd || (d >= tbFrom && xmax - xmin < tbWidth))) return false;
        /* what the sweep back will find as totalProbability, near enough: the cells of this diagonal dotted with
         * the end vector it starts from (stateMachine.c:1179-1207); cells outside the band are -inf by themselves */
        double e0, e1, e2;
#ifdef WV_VANILLA
        /* stateMachine3Vanilla_endStateProb / _raggedEndStateProb (stateMachine.c:1209-1236) */
        e0 = atEnd && it.raggedR ? (endX + endY) / 2.0 : endM;
        e1 = endX;
        e2 = endY;
#else
        if (atEnd && it.raggedR) {
            e0 = (T[T_GAP_OPEN_X] + T[T_GAP_OPEN_Y]) / 2.0;
            e1 = T[T_GAP_EXTEND_X];
            e2 = T[T_GAP_EXTEND_Y];
        } else {
            e0 = T[T_MATCH_CONTINUE];
            e1 = T[T_MATCH_FROM_GAP_X];
            e2 = T[T_MATCH_FROM_GAP_Y];
        }
#endif
        double est = CP_NEG_INF;
#pragma unroll
        for (int j = 0; j < L; j++) est = wave_fold(est, ladd(ladd(fm[j] + e0, fx[j] + e1, cf), fy[j] + e2, cf), cf);
        if (lane == 0) {
            const int from = d - (atEnd ? 0 : (int) P.tbDiags + 1);
            WvWindow w;
            w.valid = 1; w.top = d; w.from = from; w.to = tracedBackTo; w.atEnd = atEnd ? 1 : 0; w.pad = 0;
            w.est = est;
            state->win[window & 3] = w;
            state->d = d;
            state->finished = atEnd ? 1 : 0;
            state->tracedBackTo = from;
            state->cells = cells;
        }
        return true;
    };

#pragma unroll 1
    for (int db = d0 + 1; db <= D; db += feed) {
        if (db + feed > bitsHi && bitsHi < topW) {
            const int hi = topW - bitsHi > 4096 ? bitsHi + 4096 : topW;
            stage_band_steps(sh.bits, bandTab, D, bitsHi + 1, hi);
            bitsHi = hi;
        }
        {
            /* stage what this block of diagonals can ask for: the top cell's event index d - xmin - 1 and the
             * entering k-mer xmax + 1 each advance by at most one per diagonal */
            const int evTo = db + feed - 1 - xmin, rowTo = (xmax + feed + 1 <= lX + 1) ? xmax + feed + 1 : lX + 1;
#pragma unroll 1
            for (int e0 = evHi; e0 < evTo; e0 += 64) {
                const int e = e0 + lane;
                int pos = evHiMod + lane;
                pos = pos >= WV_P ? pos - WV_P : pos;
                if (e < evTo) {
                    const bool ok = e >= 0 && e < lY; /* index -1 is NULLEVENT (:261): it only ever meets -inf cells */
                    d2 v;
#ifdef WV_HDP
                    /* (grid cell, offset in it) of the event's mean, as grid_spline_interp takes them; -1 / -2 with the
                     * distance to the grid's end for a mean below / above the grid.  An event that does not exist
                     * stays at (0, 0): it only ever meets -inf cells and must score something finite */
                    v.x = 0.0;
                    v.y = 0.0;
                    if (ok) {
                        const double q = ev[3 * (long long) e];
                        if (q <= gX0) { v.x = -1.0; v.y = gX0 - q; }
                        else if (q >= gXn) { v.x = -2.0; v.y = q - gXn; }
                        else {
                            const long long il = (long long) ((q - gX0) / gDx);
                            v.x = (double) il;
                            v.y = (q - hGrid[il]) / gDx;
                        }
                    }
#elif defined(WV_VANILLA)
                    v.x = ok ? ev[3 * (long long) e] : 0.0;
                    v.y = ok ? ev[3 * (long long) e + 1] : 1.0;
                    d2 v2;
                    v2.x = ok ? 1.0 / v.y : 1.0;
                    v2.y = ok ? ev[3 * (long long) e + 2] : 0.0; /* log(noise), host libm (the batch's own copy of the
                                                                   events carries it in place of the duration) */
                    d2 *dst2 = (d2 *) sh.ev2;
                    dst2[pos] = v2;
                    dst2[pos + WV_P] = v2;
                    if (pos < L) dst2[pos + 2 * WV_P] = v2;
#else
                    v.x = ok ? ev[3 * (long long) e] : 0.0;
                    v.y = ok ? ev[3 * (long long) e + 1] : 0.0;
#endif
                    d2 *dst = (d2 *) sh.ev;
                    dst[pos] = v;
                    dst[pos + WV_P] = v;
                    if (pos < L) dst[pos + 2 * WV_P] = v;
                }
                evHiMod = (evHiMod + 64) % WV_P;
            }
            if (evTo > evHi) {
                evHiMod = ((evTo % WV_P) + WV_P) % WV_P;
                evHi = evTo;
            }
#pragma unroll 1
            for (int i = rowHi * WV_ROW + lane; i < rowTo * WV_ROW; i += 64) {
                const int x = i / WV_ROW, jj = i - x * WV_ROW;
                sh.rows[(x & (WV_ROWN - 1)) * WV_ROW + jj] = track[i];
            }
            if (rowTo > rowHi) rowHi = rowTo;
        }
        const int dbEnd = db + feed - 1 < D ? db + feed - 1 : D;
        unsigned fullMask = 0u; /* bit k: diagonal db + k keeps all three states */
#pragma unroll 1
        for (int k = 0; k < feed; k++) {
            const int dj = db + k;
            const int rHere = dj <= fromW ? rA : rB, rAbove = dj + 1 <= fromW ? rA : rB;
            fullMask |= (dj >= fullFrom || rHere == 0 || rAbove == 1 ? 1u : 0u) << k;
            rA = rA == 0 ? 9 : rA - 1;
            rB = rB == 0 ? 9 : rB - 1;
        }
#pragma unroll 1
        for (int d = db; d <= dbEnd; d += 2) {
            step(d, ((fullMask >> (d - db)) & 1u) != 0u, Am, Ax, Ay, Bm, Bx, By, RAm, RAx, RAy, RBm, RBx, RBy);
            if (finish(d, Bm, Bx, By)) return;
            step(d + 1, ((fullMask >> (d + 1 - db)) & 1u) != 0u, Bm, Bx, By, Am, Ax, Ay, RBm, RBx, RBy, RAm, RAx, RAy);
            if (finish(d + 1, Am, Ax, Ay)) return;
        }
    }
}


/* ------------------------------------------------------------------------------------------------------
 * The sweep back.
 * ------------------------------------------------------------------------------------------------------ */

/* a slot's gap-X sums (eP + tP for the three transitions into gap X) and its gap-X emission */
#ifdef WV_VANILLA
/* ... under the vanilla machine: the log transition probabilities of the slot's k-mer pair */
struct Px {
    d2 a, b, c; /* (match -> gap X, gap X -> gap X) (match -> match, gap X -> match) (match -> gap Y, -) */
};
#else
struct Px {
    d2 a, b; /* (open, extend) (switch, emission) */
};
#endif
#if defined(WV_VANILLA)
/* (WV_PX_BODY* / WV_PX_OUTS*: cpecan_wave_slots_vanilla.h) */
#elif WV_L == 1
#define WV_PX_BODY1 "ds_read_b128 %[a0], %[a]\n\t" 
#define WV_PX_OUTS1 , [a0] "+v"(p[0].a)
#define WV_PX_BODY2 "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" 
#define WV_PX_OUTS2 , [a0] "+v"(p[0].a), [a1] "+v"(p[0].b)
#elif WV_L == 2
#define WV_PX_BODY1 "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "s_branch 9f\n1:\n\t" "ds_read_b128 %[b0], %[a]\n\t" "9:\n\t" 
#define WV_PX_OUTS1 , [a0] "+v"(p[0].a), [b0] "+v"(p[1].a)
#define WV_PX_BODY2 "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "s_branch 9f\n1:\n\t" "ds_read_b128 %[b0], %[a]\n\t" "ds_read_b128 %[b1], %[a] offset:16\n\t" "9:\n\t" 
#define WV_PX_OUTS2 , [a0] "+v"(p[0].a), [a1] "+v"(p[0].b), [b0] "+v"(p[1].a), [b1] "+v"(p[1].b)
#elif WV_L == 3
#define WV_PX_BODY1 "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "s_branch 9f\n1:\n\t" "s_cmp_lg_u32 %[sel], 1\n\ts_cbranch_scc1 2f\n\t" "ds_read_b128 %[b0], %[a]\n\t" "s_branch 9f\n2:\n\t" "ds_read_b128 %[c0], %[a]\n\t" "9:\n\t" 
#define WV_PX_OUTS1 , [a0] "+v"(p[0].a), [b0] "+v"(p[1].a), [c0] "+v"(p[2].a)
#define WV_PX_BODY2 "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "s_branch 9f\n1:\n\t" "s_cmp_lg_u32 %[sel], 1\n\ts_cbranch_scc1 2f\n\t" "ds_read_b128 %[b0], %[a]\n\t" "ds_read_b128 %[b1], %[a] offset:16\n\t" "s_branch 9f\n2:\n\t" "ds_read_b128 %[c0], %[a]\n\t" "ds_read_b128 %[c1], %[a] offset:16\n\t" "9:\n\t" 
#define WV_PX_OUTS2 , [a0] "+v"(p[0].a), [a1] "+v"(p[0].b), [b0] "+v"(p[1].a), [b1] "+v"(p[1].b), [c0] "+v"(p[2].a), [c1] "+v"(p[2].b)
#elif WV_L == 4
#define WV_PX_BODY1 "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "s_branch 9f\n1:\n\t" "s_cmp_lg_u32 %[sel], 1\n\ts_cbranch_scc1 2f\n\t" "ds_read_b128 %[b0], %[a]\n\t" "s_branch 9f\n2:\n\t" "s_cmp_lg_u32 %[sel], 2\n\ts_cbranch_scc1 3f\n\t" "ds_read_b128 %[c0], %[a]\n\t" "s_branch 9f\n3:\n\t" "ds_read_b128 %[d0], %[a]\n\t" "9:\n\t" 
#define WV_PX_OUTS1 , [a0] "+v"(p[0].a), [b0] "+v"(p[1].a), [c0] "+v"(p[2].a), [d0] "+v"(p[3].a)
#define WV_PX_BODY2 "s_cmp_lg_u32 %[sel], 0\n\ts_cbranch_scc1 1f\n\t" "ds_read_b128 %[a0], %[a]\n\t" "ds_read_b128 %[a1], %[a] offset:16\n\t" "s_branch 9f\n1:\n\t" "s_cmp_lg_u32 %[sel], 1\n\ts_cbranch_scc1 2f\n\t" "ds_read_b128 %[b0], %[a]\n\t" "ds_read_b128 %[b1], %[a] offset:16\n\t" "s_branch 9f\n2:\n\t" "s_cmp_lg_u32 %[sel], 2\n\ts_cbranch_scc1 3f\n\t" "ds_read_b128 %[c0], %[a]\n\t" "ds_read_b128 %[c1], %[a] offset:16\n\t" "s_branch 9f\n3:\n\t" "ds_read_b128 %[d0], %[a]\n\t" "ds_read_b128 %[d1], %[a] offset:16\n\t" "9:\n\t" 
#define WV_PX_OUTS2 , [a0] "+v"(p[0].a), [a1] "+v"(p[0].b), [b0] "+v"(p[1].a), [b1] "+v"(p[1].b), [c0] "+v"(p[2].a), [c1] "+v"(p[2].b), [d0] "+v"(p[3].a), [d1] "+v"(p[3].b)
#endif
/* lane-masked load of a gap-X row from LDS into ONE slot (lane laneMask, layer sel: run-time, wave-uniform); the
 * row's second pair (switch sum, raw emission) only in the builds that use it */
#define WV_PX_ASM(BODY, OUTS)                                                                                     \
    unsigned long long sv;                                                                                       \
    asm volatile("s_mov_b64 %[sv], exec\n\t"                                                                      \
                 "s_mov_b64 exec, %[m]\n\t" BODY                                                                   \
                 "s_mov_b64 exec, %[sv]\n\t"                                                                      \
                 "s_waitcnt lgkmcnt(0)"                                                                            \
                 : [sv] "=&s"(sv) OUTS                                                                             \
                 : [a] "v"(rowAddr), [m] "s"(laneMask), [sel] "s"(sel)                                             \
                 : "memory", "scc");
template <bool SW> __device__ __forceinline__ void px_install(Px (&p)[WV_L], unsigned rowAddr, unsigned long long laneMask,
                                                             int sel) {
    if (SW) { WV_PX_ASM(WV_PX_BODY2, WV_PX_OUTS2) }
    else { WV_PX_ASM(WV_PX_BODY1, WV_PX_OUTS1) }
}
/* mask[sel] |= bit / &= ~bit on the scalar unit, sel a run-time value */
__device__ __forceinline__ void mask_set(unsigned long long (&mask)[WV_L], unsigned long long bit, int sel) {
#pragma unroll
    for (int j = 0; j < WV_L; j++) mask[j] |= sel == j ? bit : 0ull;
}
__device__ __forceinline__ void mask_clear(unsigned long long (&mask)[WV_L], unsigned long long bit, int sel) {
#pragma unroll
    for (int j = 0; j < WV_L; j++) mask[j] &= ~(sel == j ? bit : 0ull);
}
/* lane-masked stores of one layer's backward cells (Baum-Welch: the B ring) */
__device__ __forceinline__ void store_b3(unsigned long long laneMask, const double *rowBase, unsigned voff,
                                         double bm, double bx, double by) {
    unsigned long long sv;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_mov_b64 exec, %1\n\t"
                 "global_store_dwordx2 %2, %3, %6\n\t"
                 "global_store_dwordx2 %2, %4, %6 offset:512\n\t"
                 "global_store_dwordx2 %2, %5, %6 offset:1024\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(sv)
                 : "s"(laneMask), "v"(voff), "v"(bm), "v"(bx), "v"(by), "s"(rowBase)
                 : "memory");
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

#define WV_PXN 64 /* LDS ring of gap-X rows, by column */
struct BwdShared {
    double coef[64];
    double pxr[(WV_PXN + 1) * WV_PXW]; /* gap-X rows by column mod WV_PXN; row WV_PXN = -inf */
    unsigned bits[2][WV_BITWORDS];
    int rec[128][8]; /* phase T0: a refresh's diagonal, its band and the bands of its neighbours */
};

/* hits of a diagonal in slots before slot s (s = 0..P), from its per-layer lane masks */
__device__ __forceinline__ int hits_before(const unsigned long long (&m)[WV_L], int s) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < WV_L; j++) {
        const int q = s - j <= 0 ? 0 : (s - j + WV_L - 1) / WV_L; /* lanes l with l * L + j < s */
        n += __popcll(q >= 64 ? m[j] : m[j] & ((1ull << q) - 1ull));
    }
    return n;
}
/* rank of slot s among the hits of its diagonal in k-mer order: the band starts at slot s0 and wraps */
__device__ __forceinline__ int rank_in_diagonal(const unsigned long long (&m)[WV_L], int s0, int s) {
    const int b0 = hits_before(m, s0), b = hits_before(m, s);
    if (s >= s0) return b - b0;
    int total = 0;
#pragma unroll
    for (int j = 0; j < WV_L; j++) total += __popcll(m[j]);
    return total - b0 + b;
}

/*
 * The sweep back of one traceback window (:921-992): one anti-diagonal per iteration, cells and messages in
 * registers.  It collects the decode candidates -- cells whose F.match + B.match lies within WV_CAND_SLACK of the
 * posterior threshold measured against the forward kernel's estimate of the window's totalProbability -- and, on
 * the diagonals where the reference refreshes totalProbability, parks the backward operands of that sum in HBM
 * scratch.  The totals and the decode are the post kernel's (cpecan_k_wv_post); a window whose candidates cannot
 * be trusted is swept once more (WV_KIND_REDO) with the exact totals in hand and decoded in the loop.
 * The device selects pairs by the exponent (F+B)-total >= log(threshold) - margin; exp(), the exact
 * threshold test and floor(p * 1e7) (:776-786) are finished on the host with the reference's libm
 * (cpecan_hip.hip), so the integer posteriors are the reference's to the bit.
 */
#define WV_KIND_POSTERIOR 0 /* sweep, collect decode candidates */
#define WV_KIND_REDO 1      /* sweep once more with the exact totals in hand, pairs leave in the loop */
#define WV_KIND_EXPECT 2    /* sweep, park the backward cells for the expectation kernel */
template <bool SW, int KIND>
__device__ void backward_window(const DevItem &it, const DevParams &P, const int2 *__restrict__ bandTab,
                                const double *__restrict__ track, const double *__restrict__ model,
                                double *ring, int ringD, const WvWindow &win, ItemOut &out, BwdShared &sh,
                                WinTotal *wtot, double *vw, double *rf, int2 *candKx, double *candFb, double *bring,
                                int &nTotOut, int &nCandOut) {
    constexpr int L = WV_L;
    const int lane = threadIdx.x & 63;
    const int D = (int) (it.lX + it.lY);
    const unsigned cf = lds_addr(sh.coef);
    const unsigned pxAddr = lds_addr(sh.pxr), pxPark = pxAddr + WV_PXN * (WV_PXW * 8);
#ifdef WV_VANILLA
    const double lYM = uni64_d(model[CP_VHDR_LOG_YM]), lYY = uni64_d(model[CP_VHDR_LOG_YY]);
    const double endM = uni64_d(model[CP_VHDR_END_M]), endX = uni64_d(model[CP_VHDR_END_X]),
                 endY = uni64_d(model[CP_VHDR_END_Y]);
#endif
    const int ringMask = ringD - 1;
    double T[9];
#pragma unroll
    for (int i = 0; i < 9; i++) T[i] = model[i];

    const int dTop = uni(win.top), tracedBackFrom = uni(win.from), tracedBackTo = uni(win.to);
    const bool atEnd = uni(win.atEnd) != 0;
    const int tPost0 = dTop < tracedBackFrom ? dTop : tracedBackFrom; /* first decoded diagonal */
    const int candCap = WV_CAND_PER_DIAG * WV_L * ringD;
    if (lane < WV_PXW) sh.pxr[WV_PXN * WV_PXW + lane] = CP_NEG_INF;

    unsigned voffA[L], voffB[L]; /* byte offsets of this lane's (Fm, pm) pair and of its py inside a ring row */
#pragma unroll
    for (int j = 0; j < L; j++) {
        voffA[j] = (unsigned) ((j * WV_LAYER_DOUBLES + WV_OFF_FM(lane)) * 8);
        voffB[j] = (unsigned) (WV_ROW_PY(j, lane) * 8);
    }
    const unsigned dummyOff = (unsigned) ((long long) ringD * WV_ROW_DOUBLES * 8);
    int nTotWin = 0, nCand = 0;
    /* the forward kernel's estimate of this window's totalProbability steers the candidate test; phase T checks
     * every exact total of the window against it */
    const double totEst = uni64_d(win.est);
    const double candThr = P.logThrSlack > CP_NEG_INF ? totEst + (P.logThrSlack - WV_CAND_SLACK) : __builtin_huge_val();

    /* ------------------------------ phase S: the sweep back ------------------------------ */
    {
        /* the windows' sweeps back are the longer chain of the two that share a SIMD: they go first when both waves
         * have an instruction ready (without it: 45.7 instead of 43.9 ms per pass) */
        __builtin_amdgcn_s_setprio(WV_BACKWARD_PRIO);
        int bxmin, bxmax; /* band of the diagonal being computed */
        band_load(bandTab, dTop, bxmin, bxmax);
        /* the circular bit strings hold 8192 diagonals: longer windows are staged 4096 diagonals (whole words) at a time */
        int bitsLo = dTop - tracedBackTo > 4096 ? (dTop - 4096) & ~31 : tracedBackTo + 1;
        stage_band_steps(sh.bits, bandTab, D, bitsLo, dTop);
        /* this slot's k-mer on a diagonal: the one in (xmax - P, xmax] */
        unsigned long long fm[L]; /* the fetch cursor's lane masks */
        Px px[L];
        double Bm[L], Bx[L], By[L]; /* backward cells of the diagonal above (t+1) */
        double Um[L], Uy[L];        /* upper-block sums of t+1: By + (gap-Y emission + tP)          */
        double hB[L], hP[L];        /* B.match and match emission of t+2 (the middle block's source) */
#ifdef WV_VANILLA
        double hTm[L], hTx[L];      /* ... and that cell's log a_mm, log a_xm (its slot may change hands before they are used) */
#endif
        double pm1[L];              /* match emission of t+1 */
        {
            double e0, e1, e2; /* end state vector (stateMachine.c:1179-1207; vanilla :1209-1236) */
#ifdef WV_VANILLA
            e0 = atEnd && it.raggedR ? (endX + endY) / 2.0 : endM;
            e1 = endX;
            e2 = endY;
#else
            if (atEnd && it.raggedR) {
                e0 = (T[T_GAP_OPEN_X] + T[T_GAP_OPEN_Y]) / 2.0;
                e1 = T[T_GAP_EXTEND_X];
                e2 = T[T_GAP_EXTEND_Y];
            } else {
                e0 = T[T_MATCH_CONTINUE];
                e1 = T[T_MATCH_FROM_GAP_X];
                e2 = T[T_MATCH_FROM_GAP_Y];
            }
#endif
#pragma unroll
            for (int j = 0; j < L; j++) {
                const int sl = lane * L + j;
                int xs = sl + ((bxmin - sl + WV_P - 1) / WV_P) * WV_P;
                if (xs > bxmax) xs -= WV_P;
                const bool v = xs >= bxmin;
                fm[j] = (unsigned long long) uni64((long long) __ballot(v));
                Bm[j] = v ? e0 : CP_NEG_INF;
                Bx[j] = v ? e1 : CP_NEG_INF;
                By[j] = v ? e2 : CP_NEG_INF;
                const d2 *src = (const d2 *) (track + (long long) (v ? xs : 0) * WV_ROW + 16);
                const d2 ninf = { CP_NEG_INF, CP_NEG_INF };
                px[j].a = v ? src[0] : ninf;
                px[j].b = v ? src[1] : ninf;
#ifdef WV_VANILLA
                px[j].c = v ? src[2] : ninf;
#endif
                Um[j] = Uy[j] = hB[j] = CP_NEG_INF;
                hP[j] = pm1[j] = CP_NEG_INF;
#ifdef WV_VANILLA
                hTm[j] = hTx[j] = CP_NEG_INF;
#endif
            }
        }
        double rpo = rol1(px[0].a.x), rpe = rol1(px[0].a.y), rps = SW ? rol1(px[0].b.x) : 0.0;
#ifdef WV_VANILLA
#endif

        /* the fetch cursor: band, lane masks and slot counters of the diagonal whose ring row is fetched next.
         * A diagonal's forward match cell and two emissions are fetched WV_PREFETCH diagonals before the sweep
         * reaches it (vmcnt is one in-order queue: a fetch consumed at once would cost an HBM round trip per
         * diagonal); the loop is unrolled by the depth, so every in-flight diagonal has registers of its own.
         * A lane with no cell on that diagonal reads the dummy row (-inf): its emissions are then -inf and
         * whatever its slot holds cannot reach a cell of the band. */
        int fxmin = bxmin;
        int topL, topJ, botL, botJ; /* slots of the fetch cursor's top k-mer (next to leave) and of the k-mer below its
                                       band (next to enter) */
        {
            const int st = bxmax % WV_P, sb = (bxmin - 1 + WV_P) % WV_P;
            topL = uni(st / L); topJ = uni(st % L); /* (readfirstlane: the compiler would otherwise divide on the vector unit
                                                       and then fail to bring the masks these steer back to scalars) */
            botL = uni(sb / L); botJ = uni(sb % L);
        }
        unsigned fwMin = 0u, fwMax = 0u;
        bool fFirst = true;
        struct Rec {
            unsigned long long m[L]; /* lanes with a cell on the diagonal */
            unsigned ev;             /* what changed coming down to this diagonal: bit 0 the top k-mer left the band (its
                                        slot: lane bits 2-7, layer bits 8-9), bit 1 a k-mer entered at the bottom (lane
                                        bits 10-15, layer bits 16-17, its column mod WV_PXN bits 18-23) */
        };
        struct Q {
            double f[L], pm[L], py[L];
        };
        auto fetch = [&](const int tau, Rec &r, Q &q) __attribute__((always_inline)) {
            unsigned ev = 0u;
            if (tau > tracedBackTo && tau < dTop) {
                /* band(tau) from band(tau + 1): the steps of diagonal tau + 1 */
                const unsigned bi = (unsigned) (tau + 1) & 31u;
                if (bi == 31u || fFirst) {
                    fwMin = (unsigned) uni((int) sh.bits[0][((tau + 1) >> 5) & (WV_BITWORDS - 1)]);
                    fwMax = (unsigned) uni((int) sh.bits[1][((tau + 1) >> 5) & (WV_BITWORDS - 1)]);
                    fFirst = false;
                }
                if ((fwMax >> bi) & 1u) {
                    ev |= 1u | (unsigned) topL << 2 | (unsigned) topJ << 8;
                    mask_clear(fm, 1ull << topL, topJ);
                    topJ = uni(topJ - 1); /* (readfirstlane: see above) */
                    if (topJ < 0) { topJ = L - 1; topL = uni((topL + 63) & 63); }
                }
                if ((fwMin >> bi) & 1u) {
                    ev |= 2u | (unsigned) botL << 10 | (unsigned) botJ << 16 | (unsigned) ((fxmin - 1) & (WV_PXN - 1)) << 18;
                    mask_set(fm, 1ull << botL, botJ);
                    fxmin--;
                    botJ = uni(botJ - 1);
                    if (botJ < 0) { botJ = L - 1; botL = uni((botL + 63) & 63); }
                }
            }
            r.ev = ev;
            const bool live = tau > tracedBackTo;
            const unsigned rowOff = (unsigned) ((long long) (tau & ringMask) * (WV_ROW_DOUBLES * 8));
#pragma unroll
            for (int j = 0; j < L; j++) {
                r.m[j] = live ? fm[j] : 0ull;
                unsigned offA, offB;
                const unsigned realA = rowOff + voffA[j], dummyA = dummyOff + voffA[j];
                const unsigned realB = rowOff + voffB[j], dummyB = dummyOff + voffB[j];
                asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(offA) : "v"(dummyA), "v"(realA), "s"(r.m[j]));
                asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(offB) : "v"(dummyB), "v"(realB), "s"(r.m[j]));
                const d2 fp = *(const d2 *) ((const char *) ring + offA);
                q.f[j] = fp.x;
                q.pm[j] = fp.y;
                q.py[j] = *(const double *) ((const char *) ring + offB);
            }
        };

        Rec r0, r1;
        Q q0, q1;
        fetch(dTop, r0, q0);
        fetch(dTop - 1, r1, q1);

        int nxmin = bxmin, nxmax = bxmax; /* band of t+1 */
        int calcs = 0;
        long long emitted = 0; /* KIND_REDO: pairs written by this window so far */
        double totCur = CP_NEG_INF;

        auto step = [&](const int t, Rec &r, Q &q) __attribute__((always_inline)) {
            /* this diagonal's forward values (their loads were issued WV_PREFETCH diagonals ago), then the fetch
             * that re-uses their registers */
            double qF[L], qPm[L], qPy[L];
            unsigned long long mt[L];
#pragma unroll
            for (int j = 0; j < L; j++) { qF[j] = q.f[j]; qPm[j] = q.pm[j]; qPy[j] = q.py[j]; mt[j] = r.m[j]; }
            const unsigned ev = r.ev;
            fetch(t - WV_PREFETCH, r, q);
            if (t < dTop) {
                nxmin = bxmin; nxmax = bxmax;
                bxmax -= (int) (ev & 1u);
                bxmin -= (int) ((ev >> 1) & 1u);
                /* of slot+1: B.match and match emission of t+2 (middle block), B.gapX of t+1 with its k-mer's
                 * gap-X sums (lower block of t+1); layer L-1 takes them from layer 0 of the lane above */
                const double rhB = rol1(hB[0]), rhP = rol1(hP[0]), rBx = rol1(Bx[0]);
#ifdef WV_VANILLA
                const double rtm = rol1(hTm[0]), rtx = rol1(hTx[0]);
#endif
                /* gather form of cell_calculateBackward (:378-389): (t+2) middle block, then (t+1, smaller x-y)
                 * upper block, then (t+1, larger x-y) lower block -- the reference's scatter order per state */
                double bm[L], bx[L], by[L], y1[L], sBx[L];
#pragma unroll
                for (int j = 0; j < L; j++) {
                    const double sB = j < L - 1 ? hB[j < L - 1 ? j + 1 : 0] : rhB, sP = j < L - 1 ? hP[j < L - 1 ? j + 1 : 0] : rhP;
                    sBx[j] = j < L - 1 ? Bx[j < L - 1 ? j + 1 : 0] : rBx;
#ifdef WV_VANILLA
                    /* the transitions are those of the cell the step goes TO: slot+1's k-mer pair */
                    bm[j] = sB + (sP + (j < L - 1 ? hTm[j < L - 1 ? j + 1 : 0] : rtm));
                    bx[j] = sB + (sP + (j < L - 1 ? hTx[j < L - 1 ? j + 1 : 0] : rtx));
                    by[j] = sB + (sP + lYM);
#else
                    bm[j] = sB + (sP + T[T_MATCH_CONTINUE]);
                    bx[j] = sB + (sP + T[T_MATCH_FROM_GAP_X]);
                    by[j] = sB + (sP + T[T_MATCH_FROM_GAP_Y]);
#endif
                }
                double y2[L];
                LaddPending<L> pa, pb;
                ladd_issue<L>(pa, bm, Um, cf);
                ladd_issue<L>(pb, by, Uy, cf);
#pragma unroll
                for (int j = 0; j < L; j++) y1[j] = sBx[j] + (j < L - 1 ? px[j < L - 1 ? j + 1 : 0].a.y : rpe);
                ladd_finish<L>(pa, bm);
                ladd_issue<L>(pa, bx, y1, cf);
#pragma unroll
                for (int j = 0; j < L; j++) y2[j] = sBx[j] + (j < L - 1 ? px[j < L - 1 ? j + 1 : 0].a.x : rpo);
                ladd_finish<L>(pb, by);
                ladd_issue<L>(pb, bm, y2, cf);
                ladd_finish<L>(pa, bx);
                if (SW) {
#pragma unroll
                    for (int j = 0; j < L; j++) y1[j] = sBx[j] + (j < L - 1 ? px[j < L - 1 ? j + 1 : 0].b.x : rps);
                    ladd_issue<L>(pa, by, y1, cf);
                }
                ladd_finish<L>(pb, bm);
                if (SW) ladd_finish<L>(pa, by);
#pragma unroll
                for (int j = 0; j < L; j++) {
                    hB[j] = Bm[j]; hP[j] = pm1[j];
#ifdef WV_VANILLA
                    hTm[j] = px[j].b.x; hTx[j] = px[j].b.y; /* (before the installs below: the k-mer pairs of t+1) */
#endif
                    Bm[j] = bm[j]; Bx[j] = bx[j]; By[j] = by[j];
                }
                /* the gap-X sums above belong to the cells of t+1, the senders of the lower block: the slots of the
                 * k-mer that left the band at the top (parked) and of the one that entered at the bottom change
                 * hands only now, for the diagonals below */
                if (ev != 0u) {
                    bool touch0 = false;
                    if (ev & 1u) {
                        px_install<SW>(px, pxPark, 1ull << ((ev >> 2) & 63u), (int) ((ev >> 8) & 3u));
                        touch0 = ((ev >> 8) & 3u) == 0u;
                    }
                    if (ev & 2u) {
                        px_install<SW>(px, pxAddr + ((ev >> 18) & 63u) * (WV_PXW * 8u), 1ull << ((ev >> 10) & 63u), (int) ((ev >> 16) & 3u));
                        touch0 = touch0 || ((ev >> 16) & 3u) == 0u;
                    }
                    if (touch0) {
                        rpo = rol1(px[0].a.x);
                        rpe = rol1(px[0].a.y);
                        if (SW) rps = rol1(px[0].b.x);

                    }
                }
            }
            /* what this diagonal hands down: the upper-block sums stay in the slot */
#pragma unroll
            for (int j = 0; j < L; j++) {
#ifdef WV_VANILLA
                Um[j] = By[j] + (qPy[j] + px[j].c.x); /* (the cell above shares this slot's k-mer pair) */
                Uy[j] = By[j] + (qPy[j] + lYY);
#else
                Um[j] = By[j] + (qPy[j] + T[T_GAP_OPEN_Y]);
                Uy[j] = By[j] + (qPy[j] + T[T_GAP_EXTEND_Y]);
#endif
            }
            if (t <= tracedBackFrom) {
                double fb[L];
#pragma unroll
                for (int j = 0; j < L; j++) fb[j] = qF[j] + Bm[j];
                const int kPost = tPost0 - t;
                if (calcs++ % 10 == 0) {
                    if (KIND == WV_KIND_REDO) totCur = uni64_d(ld_agent(&wtot[kPost / 10].total));
                    else {
                        /* a refresh of totalProbability (:956-966): its per-cell terms need forward cells of two more
                         * diagonals, an HBM round trip the sweep does not wait for: the backward operands are parked
                         * in scratch -- B of this diagonal, B.match and the match emission of the one above -- and
                         * phase T0 forms the terms after the sweep */
                        double *dst = rf + (long long) nTotWin * (5 * WV_P) + lane;
#pragma unroll
                        for (int j = 0; j < L; j++) {
                            dst[(0 * L + j) * 64] = Bm[j];
                            dst[(1 * L + j) * 64] = Bx[j];
                            dst[(2 * L + j) * 64] = By[j];
                            dst[(3 * L + j) * 64] = hB[j];
                            dst[(4 * L + j) * 64] = hP[j];
                        }
                        if (lane == 0) {
                            WinTotal w;
                            w.t = t; w.xmin = bxmin; w.xmax = bxmax; w.nxmin = nxmin; w.nxmax = nxmax;
                            w.second = t + 1 <= dTop ? 1 : 0;
                            w.total = CP_NEG_INF;
                            wtot[nTotWin] = w;
                        }
                        nTotWin++;
                    }
                }
                if (KIND == WV_KIND_EXPECT) {
                    /* Baum-Welch: the backward cells go to their own ring for the expectation kernel */
                    const double *rowB = bring + (long long) (t & ringMask) * (WV_L * 3 * 64);
#pragma unroll
                    for (int j = 0; j < L; j++)
                        store_b3(mt[j], rowB, (unsigned) (j * (3 * 64 * 8) + lane * 8), Bm[j], Bx[j], By[j]);
                } else if (KIND == WV_KIND_REDO) {
                    /* the window's second sweep: exact totals are known, pairs leave in emission order */
                    unsigned long long hm[L];
                    bool hit[L];
                    const int sMin = bxmin % WV_P;
                    const int xlo = bxmin > 1 ? bxmin : 1, xhi = bxmax < t - 1 ? bxmax : t - 1;
                    unsigned long long any = 0ull;
#pragma unroll
                    for (int j = 0; j < L; j++) {
                        const int sl = lane * L + j;
                        const int x = bxmin + (sl - sMin + (sl < sMin ? WV_P : 0));
                        const double ee = fb[j] - totCur;
                        hit[j] = ((mt[j] >> lane) & 1ull) != 0ull && x >= xlo && x <= xhi && ee >= P.logThrSlack;
                        hm[j] = __ballot(hit[j]);
                        any |= hm[j];
                    }
                    if (any != 0ull) {
                        int total = 0;
#pragma unroll
                        for (int j = 0; j < L; j++) {
                            const int sl = lane * L + j;
                            if (hit[j]) {
                                const int x = bxmin + (sl - sMin + (sl < sMin ? WV_P : 0));
                                const long long idx = out.nPairs + emitted + rank_in_diagonal(hm, sMin, sl);
                                if (idx < out.pairCap) {
                                    long long *o = out.pairs + idx * 3;
                                    o[0] = 0;
                                    o[1] = x - 1;
                                    o[2] = t - x - 1;
                                    out.logp[idx] = fb[j] - totCur;
                                }
                            }
                            total += __popcll(hm[j]);
                        }
                        emitted += total;
                    }
                } else {
                    /* decode candidates: cells within WV_CAND_SLACK of the threshold against the estimate */
#pragma unroll
                    for (int j = 0; j < L; j++) {
                        const bool cand = fb[j] >= candThr;
                        const unsigned long long cm = __ballot(cand);
                        if (cm != 0ull) {
                            const int ci = nCand + __popcll(cm & ((1ull << lane) - 1ull));
                            if (cand && ci < candCap) {
                                const int sl = lane * L + j, sMin = bxmin % WV_P;
                                candKx[ci] = make_int2(kPost, bxmin + (sl - sMin + (sl < sMin ? WV_P : 0)));
                                candFb[ci] = fb[j];
                            }
                            nCand += __popcll(cm);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < L; j++) pm1[j] = qPm[j];
        };

        int t = dTop;
#pragma unroll 1
        for (;;) {
            if ((dTop - t) % 30 == 0) {
                /* gap-X rows of the k-mers that can enter during the next 30 diagonals */
                for (int i = lane; i < 32 * WV_PXW; i += 64) {
                    const int x = bxmin - 1 - i / WV_PXW, k = i % WV_PXW;
                    if (x >= 0) sh.pxr[(x & (WV_PXN - 1)) * WV_PXW + k] = track[(long long) x * WV_ROW + 16 + k];
                }
            }
            if (t - 64 < bitsLo && bitsLo > tracedBackTo + 1) {
                const int lo = bitsLo - tracedBackTo - 1 > 4096 ? bitsLo - 4096 : tracedBackTo + 1;
                stage_band_steps(sh.bits, bandTab, D, lo, bitsLo - 1);
                bitsLo = lo;
            }
            if (t <= tracedBackTo) break;
            step(t, r0, q0); t--;
            if (t <= tracedBackTo) break;
            step(t, r1, q1); t--;
        }
        if (KIND == WV_KIND_REDO) out.nPairs += emitted;
    }
    nTotOut = nTotWin;
    nCandOut = nCand;
    if (KIND == WV_KIND_REDO) return;

    /* -------------------- phase T0: the per-cell terms of every refresh -------------------- */
    /* diagonalCalculationTotalProbability (:736-754): v = cell_dotProduct(forward[t], backward[t]) (:391-397) and
     * w = matches stepping over t: forward[t-1] --match--> the cells of t+1, dotted with backward[t+1] (only the
     * match state of that clone is ever above -inf).  Lanes keep the sweep's slots.  Nothing here depends on the
     * refresh before: the records of all refreshes go to LDS first, and a refresh's operands are fetched while the
     * one before is being worked on. */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); /* the parked operands are read back below (same wave: the
                                                                   stores only have to have left; an agent-scope fence would
                                                                   write back the whole L2, the forward kernel's ring included) */
    struct Ops {
        double fm, fx, fy, s0, s1, s2, bm, bx, by, hb, hp;
#ifdef WV_VANILLA
        double tm, tx; /* log a_mm, log a_xm of the cell on t+1 */
#endif
    };
#pragma unroll 1
    for (int base = 0; base < nTotWin; base += 128) {
        const int cnt = nTotWin - base < 128 ? nTotWin - base : 128;
        for (int i = lane; i < cnt; i += 64) {
            const WinTotal *w = wtot + base + i;
            const int t = ld_agent(&w->t);
            const int2 b = bandTab[t - 1 > 0 ? t - 1 : 0];
            sh.rec[i][0] = t; sh.rec[i][1] = ld_agent(&w->xmin); sh.rec[i][2] = ld_agent(&w->xmax);
            sh.rec[i][3] = ld_agent(&w->nxmin); sh.rec[i][4] = ld_agent(&w->nxmax); sh.rec[i][5] = ld_agent(&w->second);
            sh.rec[i][6] = b.x; sh.rec[i][7] = b.y;
        }
#ifdef WV_VANILLA
        auto column_above = [&](const int i, const int j) __attribute__((always_inline)) { /* this slot's k-mer on t+1 */
            const int nmn = uni(sh.rec[i][3]);
            const int sl = lane * L + j, sMinN = nmn % WV_P;
            return nmn + (sl - sMinN + (sl < sMinN ? WV_P : 0));
        };
#endif
        auto flags = [&](const int i, const int j, bool &tv, bool &nv, bool &below) __attribute__((always_inline)) {
            const int xmn = uni(sh.rec[i][1]), xmx = uni(sh.rec[i][2]), nmn = uni(sh.rec[i][3]), nmx = uni(sh.rec[i][4]);
            const bool second = uni(sh.rec[i][5]) != 0;
            const int pxmin = uni(sh.rec[i][6]), pxmax = uni(sh.rec[i][7]);
            const int sl = lane * L + j, sMin = xmn % WV_P, sMinN = nmn % WV_P;
            const int xT = xmn + (sl - sMin + (sl < sMin ? WV_P : 0));   /* this slot's k-mer on t ... */
            const int xN = nmn + (sl - sMinN + (sl < sMinN ? WV_P : 0)); /* ... and on t+1 */
            tv = xT <= xmx;
            nv = second && xN <= nmx;
            below = nv && xN - 1 >= pxmin && xN - 1 <= pxmax;
        };
        auto issue = [&](const int i, Ops (&o)[L]) __attribute__((always_inline)) {
            const int t = uni(sh.rec[i][0]);
            const double *rowT = ring + (long long) (t & ringMask) * WV_ROW_DOUBLES;
            const double *rowB = ring + (long long) ((t - 1) & ringMask) * WV_ROW_DOUBLES;
            const double *src = rf + (long long) (base + i) * (5 * WV_P) + lane;
#pragma unroll
            for (int j = 0; j < L; j++) {
                bool tv, nv, below;
                flags(i, j, tv, nv, below);
                const int sl = lane * L + j, sb = sl == 0 ? WV_P - 1 : sl - 1; /* the slot of the k-mer below */
                const double *pa = rowT + (tv ? j : 0) * WV_LAYER_DOUBLES;
                const double *pb = rowB + (below ? sb % L : 0) * WV_LAYER_DOUBLES;
                const int la = tv ? lane : 0, lb = below ? sb / L : 0;
                o[j].fm = pa[WV_OFF_FM(la)]; o[j].fx = pa[WV_OFF_FX(la)]; o[j].fy = pa[WV_OFF_FY(la)];
                o[j].s0 = pb[WV_OFF_FM(lb)]; o[j].s1 = pb[WV_OFF_FX(lb)]; o[j].s2 = pb[WV_OFF_FY(lb)];
                /* (plain loads: these lines were written by this wave, behind a release fence, and never read before) */
                o[j].bm = src[(0 * L + j) * 64];
                o[j].bx = src[(1 * L + j) * 64];
                o[j].by = src[(2 * L + j) * 64];
                o[j].hb = src[(3 * L + j) * 64];
                o[j].hp = src[(4 * L + j) * 64];
#ifdef WV_VANILLA
                const double *tr = track + (long long) (nv ? column_above(i, j) : 0) * WV_ROW;
                o[j].tm = tr[18];
                o[j].tx = tr[19];
#endif
            }
        };
        auto work = [&](const int i, const Ops (&o)[L]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < L; j++) {
                bool tv, nv, below;
                flags(i, j, tv, nv, below);
                if (tv) {
                    double v = o[j].fm + o[j].bm;
                    v = ladd(v, o[j].fx + o[j].bx, cf);
                    v = ladd(v, o[j].fy + o[j].by, cf);
                    vw[((long long) (base + i) * 2 + 0) * WV_P + lane * L + j] = v;
                }
                if (nv) {
                    const double s0 = below ? o[j].s0 : CP_NEG_INF, s1 = below ? o[j].s1 : CP_NEG_INF,
                                 s2 = below ? o[j].s2 : CP_NEG_INF;
#ifdef WV_VANILLA
                    double mm = s0 + (o[j].hp + o[j].tm);
                    mm = ladd(mm, s1 + (o[j].hp + o[j].tx), cf);
                    mm = ladd(mm, s2 + (o[j].hp + lYM), cf);
#else
                    double mm = s0 + (o[j].hp + T[T_MATCH_CONTINUE]);
                    mm = ladd(mm, s1 + (o[j].hp + T[T_MATCH_FROM_GAP_X]), cf);
                    mm = ladd(mm, s2 + (o[j].hp + T[T_MATCH_FROM_GAP_Y]), cf);
#endif
                    vw[((long long) (base + i) * 2 + 1) * WV_P + lane * L + j] = mm + o[j].hb;
                }
            }
        };
        Ops oa[L], ob[L];
        issue(0, oa);
#pragma unroll 1
        for (int i = 0; i < cnt; i += 2) {
            if (i + 1 < cnt) issue(i + 1, ob);
            work(i, oa);
            if (i + 2 < cnt) issue(i + 2, oa);
            if (i + 1 < cnt) work(i + 1, ob);
        }
    }
}

} // namespace

/* One wave per alignment: forward sweep up to its next traceback point.  Two instantiations: a -inf
 * gapY->gapX transition (the nanopore default, stateMachine.c:1287) contributes logAdd(acc, -inf) == acc, and the
 * build without that term is the one a batch runs on when none of its models has the transition. */
template <bool SW> __device__ __forceinline__ void wv_forward_kernel(
    const DevItem *__restrict__ items, long long nItems, const DevParams &P,
    const int2 *__restrict__ bandTab, const double *__restrict__ track,
    const long long *__restrict__ trackBase, const double *__restrict__ events,
    const double *__restrict__ models, double *Fring, long long ringDoubles, int ringD,
    WvState *states, int window, FwdShared (&shs)[WV_WPB]) {
    const long long idx = (long long) blockIdx.x * WV_WPB + uni(threadIdx.x >> 6);
    FwdShared &sh = shs[uni(threadIdx.x >> 6)];
    if (idx >= nItems) return;
    WvState *state = states + idx;
    const DevItem it = uniform_item(items[idx]);
    if (state->finished || it.lX + it.lY == 0) return;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    init_coef(sh.coef);
    forward_window<SW>(it, P, bandTab + it.diagBase, track + trackBase[idx] * WV_ROW, events,
                       models + (long long) it.model * WV_MODEL_DOUBLES, Fring + idx * ringDoubles, ringD, state, window,
                       sh);
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        state->clkShader += (long long) (c1 - c0);
        state->clkRef += (long long) (r1 - r0);
    }
}
extern "C" __global__ __launch_bounds__(64 * WV_WPB) void WV_SYM(cpecan_k_wv_forward)(
    const DevItem *__restrict__ items, long long nItems, DevParams P,
    const int2 *__restrict__ bandTab, const double *__restrict__ track,
    const long long *__restrict__ trackBase, const double *__restrict__ events,
    const double *__restrict__ models, double *Fring, long long ringDoubles, int ringD, WvState *states, int window) {
    __shared__ FwdShared sh[WV_WPB];
    wv_forward_kernel<false>(items, nItems, P, bandTab, track, trackBase, events, models, Fring, ringDoubles, ringD,
                             states, window, sh);
}
extern "C" __global__ __launch_bounds__(64 * WV_WPB) void WV_SYM(cpecan_k_wv_forward_sw)(
    const DevItem *__restrict__ items, long long nItems, DevParams P,
    const int2 *__restrict__ bandTab, const double *__restrict__ track,
    const long long *__restrict__ trackBase, const double *__restrict__ events,
    const double *__restrict__ models, double *Fring, long long ringDoubles, int ringD, WvState *states, int window) {
    __shared__ FwdShared sh[WV_WPB];
    wv_forward_kernel<true>(items, nItems, P, bandTab, track, trackBase, events, models, Fring, ringDoubles, ringD,
                            states, window, sh);
}

/* One wave per alignment: the sweep back of the window the forward kernel just described (win[].valid: see
 * cpecan_sweep.h).  What follows the sweep -- the refresh terms, the totals, the decode -- is the post kernel's. */
template <bool SW, int KIND> __device__ __forceinline__ void wv_backward_kernel(
    const DevItem *__restrict__ items, long long nItems, const DevParams &P,
    const int2 *__restrict__ bandTab, const double *__restrict__ track,
    const long long *__restrict__ trackBase, const double *__restrict__ models, double *Fring,
    long long ringDoubles, int ringD, WvState *states, long long *pairs, double *pairLogp,
    char *scratch, long long scratchBytes, double *Bring, int window, BwdShared (&shs)[WV_WPB]) {
    const long long idx = (long long) blockIdx.x * WV_WPB + uni(threadIdx.x >> 6);
    BwdShared &sh = shs[uni(threadIdx.x >> 6)];
    if (idx >= nItems) return;
    WvState *state = states + idx;
    WvWindow win;
    {
        const WvWindow *w = &state->win[window & 3];
        win.valid = ld_agent(&w->valid); win.top = ld_agent(&w->top); win.from = ld_agent(&w->from);
        win.to = ld_agent(&w->to); win.atEnd = ld_agent(&w->atEnd); win.nCand = win.nRefresh = win.pad = 0;
        win.est = ld_agent(&w->est);
    }
    if (uni(win.valid) != (KIND == WV_KIND_REDO ? 2 : 1)) return;
    const DevItem it = uniform_item(items[idx]);
    init_coef(sh.coef);
    ItemOut out;
    out.pairs = pairs + it.pairBase * 3;
    out.logp = pairLogp + it.pairBase;
    out.pairCap = it.pairCap;
    out.totXay = nullptr;
    out.totVal = nullptr;
    out.totCap = 0;
    out.nPairs = uni64(ld_agent(&state->nPairs));
    out.nTot = 0;
    char *sc = scratch + idx * scratchBytes;
    const long long nW = (long long) ringD / 10 + 8;
    WinTotal *wtot = (WinTotal *) (sc + 2ll * ringD * sizeof(int));
    double *vw = (double *) (sc + 2ll * ringD * sizeof(int) + nW * sizeof(WinTotal));
    double *rf = vw + nW * 2 * WV_P;
    unsigned long long *msk = (unsigned long long *) (rf + nW * 5 * WV_P);
    int2 *candKx = (int2 *) ((char *) msk + 4ll * ringD * sizeof(unsigned long long));
    double *candFb = (double *) ((char *) candKx + (long long) WV_L * WV_CAND_PER_DIAG * ringD * sizeof(int2));
    int nTot = 0, nCand = 0;
    backward_window<SW, KIND>(it, P, bandTab + it.diagBase, track + trackBase[idx] * WV_ROW,
                              models + (long long) it.model * WV_MODEL_DOUBLES, Fring + idx * ringDoubles, ringD, win,
                              out, sh, wtot, vw, rf, candKx, candFb,
                              Bring ? Bring + idx * ((long long) ringD * WV_L * 3 * 64) : nullptr, nTot, nCand);
    if ((threadIdx.x & 63) == 0) {
        WvWindow *w = &state->win[window & 3];
        if (KIND == WV_KIND_REDO) {
            state->nPairs = out.nPairs;
            w->valid = 0;
        } else {
            w->nCand = nCand;
            w->nRefresh = nTot;
            w->valid = 3;
        }
    }
}
#define WV_BACKWARD_KERNEL(name, SW, KIND)                                                                        \
    extern "C" __global__ __launch_bounds__(64 * WV_WPB) void WV_SYM(name)(                                       \
        const DevItem *__restrict__ items, long long nItems, DevParams P, const int2 *__restrict__ bandTab,       \
        const double *__restrict__ track, const long long *__restrict__ trackBase,                                \
        const double *__restrict__ models, double *Fring, long long ringDoubles, int ringD, WvState *states,      \
        long long *pairs, double *pairLogp, char *scratch, long long scratchBytes, double *Bring, int window) {   \
        __shared__ BwdShared sh[WV_WPB];                                                                          \
        wv_backward_kernel<SW, KIND>(items, nItems, P, bandTab, track, trackBase, models, Fring, ringDoubles,     \
                                     ringD, states, pairs, pairLogp, scratch, scratchBytes, Bring, window, sh);   \
    }
WV_BACKWARD_KERNEL(cpecan_k_wv_backward, false, WV_KIND_POSTERIOR)
WV_BACKWARD_KERNEL(cpecan_k_wv_backward_sw, true, WV_KIND_POSTERIOR)
WV_BACKWARD_KERNEL(cpecan_k_wv_resweep, false, WV_KIND_REDO)
WV_BACKWARD_KERNEL(cpecan_k_wv_resweep_sw, true, WV_KIND_REDO)
WV_BACKWARD_KERNEL(cpecan_k_wv_backward_em, false, WV_KIND_EXPECT)
#if !defined(WV_VANILLA) /* (no gap Y -> gap X transition in the vanilla machine) */
WV_BACKWARD_KERNEL(cpecan_k_wv_backward_em_sw, true, WV_KIND_EXPECT)
#endif

/*
 * What follows a sweep back, one 256-thread workgroup per alignment (none of it is a recurrence along the
 * diagonals, so it is spread over four waves and over many resident workgroups instead of waiting inside the
 * sweep's single wave):
 *  T  the totals of the window's totalProbability refreshes (their per-cell terms were formed by the sweep's wave,
 *     phase T0): the reference's order-dependent logAdd fold is inherently serial, so each THREAD folds one
 *     diagonal's terms privately, in the reference's order (dpDiagonal_dotProduct :587-597);
 *  D  diagonalCalculationPosteriorMatchProbs (:756-795) from the sweep's candidate list: hits are marked per
 *     diagonal, prefix-summed in emission order (diagonals descending, x-y ascending) and written.  A window
 *     whose candidates cannot be trusted (a total strays from the forward kernel's estimate, the list overflowed,
 *     threshold 0) is left to the re-sweep kernel.
 */
struct PostShared {
    double coef[64];
    double vbuf[256];
    int part[4];
    int scan, carry;
};
extern "C" __global__ __launch_bounds__(256) void WV_SYM(cpecan_k_wv_post)(
    const DevItem *__restrict__ items, long long nItems, DevParams P, const int2 *__restrict__ bandTabAll,
    const double *__restrict__ models, const double *Fring, long long ringDoubles, int ringD, WvState *states,
    long long *pairs, double *pairLogp, long long *totXay, double *totVal, char *scratch, long long scratchBytes,
    int window, int formTerms) {
    constexpr int L = WV_L;
    __shared__ PostShared sh;
    const long long idx = blockIdx.x;
    if (idx >= nItems) return;
    WvState *state = states + idx;
    WvWindow *wp = &state->win[window & 3];
    if (uni(ld_agent(&wp->valid)) != 3) return;
    const DevItem it = uniform_item(items[idx]);
    const int tid = threadIdx.x, lane = tid & 63, wv = uni(tid >> 6);
    if (tid < 64) init_coef(sh.coef);
    const int dTop = uni(ld_agent(&wp->top)), tracedBackFrom = uni(ld_agent(&wp->from)), tracedBackTo = uni(ld_agent(&wp->to));
    const int nTotWin = uni(ld_agent(&wp->nRefresh)), nCand = uni(ld_agent(&wp->nCand));
    const double totEst = uni64_d(ld_agent(&wp->est));
    const int tPost0 = dTop < tracedBackFrom ? dTop : tracedBackFrom; /* first decoded diagonal */
    const int nPost = tPost0 - tracedBackTo;                         /* diagonals decoded      */
    const int candCap = WV_CAND_PER_DIAG * WV_L * ringD;
    if (tid == 0) sh.scan = (P.scanDecode != 0 || !(P.logThrSlack > CP_NEG_INF) || nCand > candCap) ? 1 : 0;
    __syncthreads();
    const unsigned cf = lds_addr(sh.coef);
    const int2 *bandTab = bandTabAll + it.diagBase;
    char *sc = scratch + idx * scratchBytes;
    const long long nW = (long long) ringD / 10 + 8;
    int *offBuf = (int *) sc;
    WinTotal *wtot = (WinTotal *) (sc + 2ll * ringD * sizeof(int));
    double *vw = (double *) (sc + 2ll * ringD * sizeof(int) + nW * sizeof(WinTotal));
    const double *rf = vw + nW * 2 * WV_P;
    unsigned long long *msk = (unsigned long long *) (rf + nW * 5 * WV_P);
    const int2 *candKx = (const int2 *) ((char *) msk + 4ll * ringD * sizeof(unsigned long long));
    const double *candFb = (const double *) ((const char *) candKx + (long long) WV_L * WV_CAND_PER_DIAG * ringD * sizeof(int2));
    long long nPairs0 = uni64(ld_agent(&state->nPairs)), nTot0 = uni64(ld_agent(&state->nTot));

    (void) models; (void) Fring; (void) ringDoubles; (void) formTerms;

    /* ------------------------------ phase T: the totals ------------------------------ */
    /* (plain loads throughout: what this kernel reads was written by an earlier kernel, or by this workgroup
     * before a __syncthreads(); atomic loads would each be waited for on their own) */
#pragma unroll 1
    for (int k0 = 0; k0 < 2 * nTotWin; k0 += 256) {
        const int k = k0 + tid;
        double acc = CP_NEG_INF;
        WinTotal w;
        w.second = 0; w.t = 0; w.xmin = w.xmax = w.nxmin = w.nxmax = 0;
        const int f = k & 1;
        if (k < 2 * nTotWin) {
            w = wtot[k >> 1];
            if (f == 0 || w.second) {
                const int lo = f ? w.nxmin : w.xmin, hi = f ? w.nxmax : w.xmax;
                const double *src = vw + ((long long) (k >> 1) * 2 + f) * WV_P;
                double v[8], nv[8]; /* the next eight terms are in flight while these eight are folded */
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = lo + q <= hi ? src[(lo + q) % WV_P] : CP_NEG_INF;
#pragma unroll 1
                for (int x0 = lo; x0 <= hi; x0 += 8) {
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        nv[q] = x0 + 8 + q <= hi ? src[(x0 + 8 + q) % WV_P] : CP_NEG_INF;
#pragma unroll
                    for (int q = 0; q < 8; q++) acc = ladd(acc, v[q], cf);
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = nv[q];
                }
            }
        }
        sh.vbuf[tid] = acc;
        __syncthreads();
        const double partner = sh.vbuf[(tid + 1) & 255];
        if (k < 2 * nTotWin && f == 0) {
            double tot = acc;
            if (w.second) tot = ladd(acc, partner, cf);
            wtot[k >> 1].total = tot;
            if (!(fabs(tot - totEst) <= WV_CAND_SLACK)) sh.scan = 1; /* also catches NaN and infinities */
            const long long o = nTot0 + (k >> 1);
            if (o < it.totCap) {
                totXay[it.totBase + o] = w.t;
                totVal[it.totBase + o] = tot;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (P.mode != 0 || nPost <= 0 || sh.scan != 0) {
        if (tid == 0) {
            state->nTot = nTot0 + nTotWin;
            /* Baum-Welch: nothing to decode; otherwise a window whose candidates cannot be trusted goes to the
             * re-sweep kernel (the exact totals are in scratch by now) */
            wp->valid = (P.mode == 0 && nPost > 0) ? 2 : 0;
            if (P.mode != 0) state->expectPending = window + 1; /* which launch's window the B ring holds */
        }
        return;
    }

    /* ------------------------------ phase D: the aligned pairs ------------------------------ */
    long long *outPairs = pairs + it.pairBase * 3;
    double *outLogp = pairLogp + it.pairBase;
    for (int i = tid; i < nPost * 4; i += 256) msk[i] = 0ull;
    __syncthreads();
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll 1
        for (int i = tid; i < nCand; i += 256) {
            const int2 kx = candKx[i];
            const int k = kx.x, x = kx.y, t = tPost0 - k;
            const double ee = candFb[i] - wtot[k / 10].total;
            if (!(x >= 1 && x <= t - 1 && ee >= P.logThrSlack)) continue;
            const int sl = x % WV_P;
            if (!pass) {
                atomicOr(msk + k * 4ll + sl % L, 1ull << (sl / L));
                continue;
            }
            unsigned long long mm[L];
#pragma unroll
            for (int j = 0; j < L; j++) mm[j] = msk[k * 4ll + j];
            const int rank = rank_in_diagonal(mm, bandTab[t].x % WV_P, sl);
            const long long o = nPairs0 + offBuf[k] + rank;
            if (o < it.pairCap) {
                long long *dst = outPairs + o * 3;
                dst[0] = 0;
                dst[1] = x - 1;
                dst[2] = t - x - 1;
                outLogp[o] = ee;
            }
        }
            __syncthreads();
        if (!pass) {
            /* hits per diagonal from the masks; exclusive prefix in emission order */
            int carry = 0;
#pragma unroll 1
            for (int base = 0; base < nPost; base += 8 * 256) {
                const int b0 = base + tid * 8;
                int h[8], sum = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    h[q] = 0;
                    if (b0 + q < nPost) {
#pragma unroll
                        for (int j = 0; j < L; j++) h[q] += __popcll(msk[(b0 + q) * 4ll + j]);
                    }
                    sum += h[q];
                }
                int inc = sum;
#pragma unroll
                for (int o2 = 1; o2 < 64; o2 <<= 1) {
                    const int up = __shfl_up(inc, o2);
                    if (lane >= o2) inc += up;
                }
                __syncthreads(); /* part[] of the previous round has been read */
                if (lane == 63) sh.part[wv] = inc;
                __syncthreads();
                int o = carry + inc - sum;
#pragma unroll
                for (int w2 = 0; w2 < 4; w2++) {
                    const int c = sh.part[w2];
                    if (w2 < wv) o += c;
                    carry += c;
                }
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (b0 + q < nPost) {
                        offBuf[b0 + q] = o;
                        o += h[q];
                    }
            }
            if (tid == 0) sh.carry = carry;
                    __syncthreads();
        }
    }
    if (tid == 0) {
        state->nPairs = nPairs0 + sh.carry;
        state->nTot = nTot0 + nTotWin;
        wp->valid = 0;
    }
}

#if defined(WV_VANILLA)
/*
 * The vanilla machine's expectations of the traceback window the backward kernel just swept
 * (diagonalCalculation_Expectations :841-863 with cell_signal_updateBetaAndAlphaProb :478-498): of all transitions only
 * match -> gap X (into the skip bin of the cell's k-mer pair) and gap X -> gap X (bin + 30) are collected, both from the
 * cell's lower neighbour; the gap-X emission of this machine is 0.  Element-wise over the forward ring, the B ring and
 * the window's exact totals, as the strawMan pass below: 64 * L threads, wave j takes layer j.  A thread keeps the two
 * sums of its column in registers and adds them to the column's bin (track row entry 21) when its slot moves on.
 */
extern "C" __global__ __launch_bounds__(WV_P) void WV_SYM(cpecan_k_wv_expect)(
    const DevItem *__restrict__ items, long long nItems, DevParams P, const int2 *__restrict__ bandTab,
    const double *__restrict__ track, const long long *__restrict__ trackBase,
    const unsigned short *__restrict__ kidx, const double *__restrict__ models, const double *Fring,
    long long ringDoubles, const double *Bring, int ringD, WvState *states, const char *scratch,
    long long scratchBytes, double *expect, int window, long long *pairs, double *pairLogp) {
    constexpr int L = WV_L;
    __shared__ double sBins[64];
    (void) P; (void) kidx; (void) models; (void) pairs; (void) pairLogp;
    const long long idx = blockIdx.x;
    if (idx >= nItems) return;
    const WvState *state = states + idx;
    if (state->expectPending != window + 1) return;
    const DevItem it = uniform_item(items[idx]);
    const int lane = threadIdx.x & 63, j = uni(threadIdx.x >> 6);
    const int sl = lane * L + j, sb = sl == 0 ? WV_P - 1 : sl - 1; /* this thread's slot and the one below it */
    const int ringMask = ringD - 1;
    const double *ring = Fring + idx * ringDoubles;
    const double *blw = ring + (sb % L) * WV_LAYER_DOUBLES;     /* the slot below: layer sb % L, lane sb / L */
    const int lb = sb / L;
    const double *bown = Bring + idx * ((long long) ringD * L * 3 * 64) + j * (3 * 64) + lane;
    const double *tr = track + trackBase[idx] * WV_ROW;
    const int2 *tab = bandTab + it.diagBase;
    const WinTotal *wtot = (const WinTotal *) (scratch + idx * scratchBytes + 2ll * ringD * sizeof(int));
    const int dTop = uni(state->win[window & 3].top), from = uni(state->win[window & 3].from),
              to = uni(state->win[window & 3].to);
    const int tPost0 = dTop < from ? dTop : from;
    double *dst = expect + (long long) it.model * (60 + 1);
    if (threadIdx.x < 64) sBins[threadIdx.x] = 0.0;
    __syncthreads();

    double lik = 0.0, beta = 0.0, alpha = 0.0;
    int col = -1; /* matrix column whose sums beta / alpha hold */
    const int perChunk = (tPost0 - to + (int) gridDim.y - 1) / (int) gridDim.y;
    const int tHi = tPost0 - (int) blockIdx.y * perChunk;             /* this workgroup: diagonals tHi .. tLo+1 */
    const int tLo = tHi - perChunk > to ? tHi - perChunk : to;
    if (tHi > to) {
        int b0min, b0max, b1min, b1max;
        band_load(tab, tHi, b0min, b0max);
        int xs = sl + ((b0min - sl + WV_P - 1) / WV_P) * WV_P; /* this slot's k-mer: the one in (xmax-P, xmax] */
        if (xs > b0max) xs -= WV_P;
#pragma unroll 1
        for (int t = tHi; t > tLo; t--) {
            band_load(tab, t - 1, b1min, b1max);
            if (xs > b0max) xs -= WV_P;
            const int x = xs;
            const double total = wtot[(tPost0 - t) / 10].total;
            if (threadIdx.x == 0) lik += total;
            if (x >= b0min && x - 1 >= b1min && x - 1 <= b1max) { /* the cell (t, x) and its lower neighbour exist */
                const long long r1 = (long long) ((t - 1) & ringMask) * WV_ROW_DOUBLES;
                const double Bx = bown[(long long) (t & ringMask) * (L * 3 * 64) + 64];
                const double l0 = blw[r1 + WV_OFF_FM(lb)], l1 = blw[r1 + WV_OFF_FX(lb)];
                const double *row = tr + (long long) x * WV_ROW;
                if (x != col) {
                    if (col >= 0) {
                        const int bin = (int) tr[(long long) col * WV_ROW + 21];
                        atomicAdd(&sBins[bin], beta);
                        atomicAdd(&sBins[bin + 30], alpha);
                    }
                    col = x;
                    beta = alpha = 0.0;
                }
                beta += exp(l0 + Bx + (0 + row[16]) - total);
                alpha += exp(l1 + Bx + (0 + row[17]) - total);
            }
            b0min = b1min; b0max = b1max;
        }
        if (col >= 0) {
            const int bin = (int) tr[(long long) col * WV_ROW + 21];
            atomicAdd(&sBins[bin], beta);
            atomicAdd(&sBins[bin + 30], alpha);
        }
    }
    __syncthreads();
    if (threadIdx.x < 60 && sBins[threadIdx.x] != 0.0) atomicAdd(dst + threadIdx.x, sBins[threadIdx.x]);
    if (threadIdx.x == 0 && lik != 0.0) atomicAdd(dst + 60, lik);
}
#endif

#if !defined(WV_VANILLA)
/*
 * Baum-Welch expectations of the traceback window the backward kernel just swept
 * (diagonalCalculation_Expectations :841-863 with cell_signal_updateTransAndKmerSkipExpectations :426-443).
 * By now every operand is in HBM -- forward cells and the two event-dependent emissions in the forward ring,
 * backward cells in the B ring, the window's exact totals in scratch -- so this is an element-wise pass
 * with no recurrence: per cell eight exp(F.from + B.to + (eP + tP) - total), summed per thread and reduced
 * once per window.  64 * L threads: wave j takes layer j of the ring rows, lane by lane.  A thread keeps the
 * sum of its k-mer's gap-X expectations in a register and adds it to the k-mer's bin when its slot moves to
 * another k-mer.  The match block is skipped where forward[t-2] has been freed by then, as in the reference.
 *
 * The HDP machine (-DWV_HDP; cell_signal_updateTransAndKmerSkipExpectations2 :445-476) collects the nine transitions
 * and the likelihood the same way, no k-mer bins, and an ASSIGNMENT for every transition into match whose own
 * posterior reaches the threshold: (from-state + 4 * window, x, y) with its exponent, appended to the alignment's pair
 * list in whatever order the threads get there -- the host puts them into the reference's order (windows ascending,
 * diagonals descending, x ascending, from-state ascending) when it fetches them.
 */
extern "C" __global__ __launch_bounds__(WV_P) void WV_SYM(cpecan_k_wv_expect)(
    const DevItem *__restrict__ items, long long nItems, DevParams P, const int2 *__restrict__ bandTab,
    const double *__restrict__ track, const long long *__restrict__ trackBase,
    const unsigned short *__restrict__ kidx, const double *__restrict__ models, const double *Fring,
    long long ringDoubles, const double *Bring, int ringD, WvState *states, const char *scratch,
    long long scratchBytes, double *expect, int window, long long *pairs, double *pairLogp) {
    constexpr int L = WV_L;
    __shared__ double sExp[16];
    const long long idx = blockIdx.x;
    if (idx >= nItems) return;
    const WvState *state = states + idx;
    if (state->expectPending != window + 1) return;
    const DevItem it = uniform_item(items[idx]);
    const int lane = threadIdx.x & 63, j = uni(threadIdx.x >> 6);
    const int sl = lane * L + j, sb = sl == 0 ? WV_P - 1 : sl - 1; /* this thread's slot and the one below it */
    const int ringMask = ringD - 1;
    const double *ring = Fring + idx * ringDoubles;
    const double *own = ring + j * WV_LAYER_DOUBLES;            /* this thread's layer of a ring row, its lane `lane` */
    const double *blw = ring + (sb % L) * WV_LAYER_DOUBLES;     /* the slot below: layer sb % L, lane sb / L */
    const int lb = sb / L;
    const double *bown = Bring + idx * ((long long) ringD * L * 3 * 64) + j * (3 * 64) + lane;
    const double *tr = track + trackBase[idx] * WV_ROW;
    const unsigned short *kx = kidx + it.xOff;
    const int2 *tab = bandTab + it.diagBase;
    const WinTotal *wtot = (const WinTotal *) (scratch + idx * scratchBytes + 2ll * ringD * sizeof(int));
    const int dTop = uni(state->win[window & 3].top), from = uni(state->win[window & 3].from),
              to = uni(state->win[window & 3].to);
    const int tPost0 = dTop < from ? dTop : from;
#ifdef WV_HDP
    double *dst = expect + (long long) it.model * (9 + 1);
    const double *tm = ((const DevHdpModel *) models)[it.model].t;
    WvState *stateW = states + idx;
#else
    double *dst = expect + (long long) it.model * (9 + 4096 + 1);
    const double *tm = models + (long long) it.model * CP_MODEL_STRIDE;
#endif

    double acc[8]; /* M>X X>X Y>X | M>M X>M Y>M | M>Y Y>Y */
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = 0.0;
    double lik = 0.0, gapSum = 0.0;
    int gapX = -1; /* matrix column whose gap-X expectations gapSum holds */

    const int perChunk = (tPost0 - to + (int) gridDim.y - 1) / (int) gridDim.y;
    const int tHi = tPost0 - (int) blockIdx.y * perChunk;             /* this workgroup: diagonals tHi .. tLo+1 */
    const int tLo = tHi - perChunk > to ? tHi - perChunk : to;
    if (tHi <= to) return;
    int b0min, b0max, b1min, b1max, b2min, b2max;
    band_load(tab, tHi, b0min, b0max);
    band_load(tab, tHi - 1, b1min, b1max);
    int xs = sl + ((b0min - sl + WV_P - 1) / WV_P) * WV_P; /* this slot's k-mer: the one in (xmax-P, xmax] */
    if (xs > b0max) xs -= WV_P;
#pragma unroll 1
    for (int t = tHi; t > tLo; t--) {
        band_load(tab, t - 2, b2min, b2max);
        if (xs > b0max) xs -= WV_P;
        const int x = xs;
        const double total = wtot[(tPost0 - t) / 10].total;
        if (threadIdx.x == 0) lik += total;
        if (x >= b0min) { /* the cell (t, x) exists */
            const long long rt = (long long) (t & ringMask) * WV_ROW_DOUBLES, r1 = (long long) ((t - 1) & ringMask) * WV_ROW_DOUBLES,
                            r2 = (long long) ((t - 2) & ringMask) * WV_ROW_DOUBLES;
            const double *bc = bown + (long long) (t & ringMask) * (L * 3 * 64);
            const double Bm = bc[0], Bx = bc[64], By = bc[128];
            const bool vLower = x - 1 >= b1min && x - 1 <= b1max;
            const bool vMiddle = t - 2 >= to && x - 1 >= b2min && x - 1 <= b2max;
            const bool vUpper = x >= b1min && x <= b1max;
            if (vLower) {
                const double l0 = blw[r1 + WV_OFF_FM(lb)], l1 = blw[r1 + WV_OFF_FX(lb)], l2 = blw[r1 + WV_OFF_FY(lb)];
                const double *row = tr + (long long) x * WV_ROW;
                const double p0 = exp(l0 + Bx + row[16] - total);
                const double p1 = exp(l1 + Bx + row[17] - total);
                const double p2 = exp(l2 + Bx + row[18] - total);
                acc[0] += p0;
                acc[1] += p1;
                acc[2] += p2;
#ifndef WV_HDP
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
#endif
            }
            if (vMiddle) {
                const double m0 = blw[r2 + WV_OFF_FM(lb)], m1 = blw[r2 + WV_OFF_FX(lb)], m2 = blw[r2 + WV_OFF_FY(lb)];
                const double eP = own[rt + WV_OFF_PM(lane)];
                const double e0 = m0 + Bm + (eP + tm[T_MATCH_CONTINUE]) - total;
                const double e1 = m1 + Bm + (eP + tm[T_MATCH_FROM_GAP_X]) - total;
                const double e2 = m2 + Bm + (eP + tm[T_MATCH_FROM_GAP_Y]) - total;
                const double q0 = exp(e0), q1 = exp(e1), q2 = exp(e2);
                acc[3] += q0;
                acc[4] += q1;
                acc[5] += q2;
#ifdef WV_HDP
                const double ee[3] = { e0, e1, e2 }, qq[3] = { q0, q1, q2 };
#pragma unroll
                for (int f = 0; f < 3; f++)
                    if (qq[f] >= P.threshold) {
                        const long long at = (long long) atomicAdd((unsigned long long *) &stateW->nPairs, 1ull);
                        if (at < it.pairCap) {
                            long long *o = pairs + (it.pairBase + at) * 3;
                            o[0] = f + 4ll * window;
                            o[1] = x - 1;
                            o[2] = (t - x) - 1;
                            pairLogp[it.pairBase + at] = ee[f];
                        }
                    }
#endif
            }
            if (vUpper) {
                const double u0 = own[r1 + WV_OFF_FM(lane)], u2 = own[r1 + WV_OFF_FY(lane)];
                const double eP = ring[rt + WV_ROW_PY(j, lane)];
                acc[6] += exp(u0 + By + (eP + tm[T_GAP_OPEN_Y]) - total);
                acc[7] += exp(u2 + By + (eP + tm[T_GAP_EXTEND_Y]) - total);
            }
        }
        b0min = b1min; b0max = b1max;
        b1min = b2min; b1max = b2max;
    }
#ifndef WV_HDP
    if (gapX > 0) {
        const int k = kx[gapX - 1];
        if (k < 4096) atomicAdd(dst + 9 + k, gapSum);
    }
#else
    (void) gapX; (void) gapSum; (void) kx;
#endif
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
#ifdef WV_HDP
    if (threadIdx.x == 0) atomicAdd(dst + 9, lik);
#else
    if (threadIdx.x == 0) atomicAdd(dst + 9 + 4096, lik);
#endif
}

#endif /* strawMan and HDP builds */

#if WV_L == 4 && !defined(WV_HDP) && !defined(WV_VANILLA)
/* per-item track of emission constants, wave layout: column x (0..lX) = the 16 emission constants of the k-mer
 * that matrix column x scores (column 0 = the "not a k-mer" sentinel, sequence_getKmer index -1, :314-318),
 * its gap-X emission plus each of the three transitions into gap X (the eP + tP of cell_calculate*), and the
 * emission itself */
extern "C" __global__ void cpecan_k_wv_track(const DevItem *__restrict__ items, long long nItems,
                                             const long long *__restrict__ trackBase,
                                             const unsigned short *__restrict__ kidx,
                                             const double *__restrict__ models, double *track) {
    const long long item = blockIdx.y;
    if (item >= nItems) return;
    const DevItem it = items[item];
    const double *model = models + (long long) it.model * CP_MODEL_STRIDE;
    const double *rows = model + CP_MODEL_HEADER;
    const long long n = (it.lX + 1) * WV_ROW;
    double *dst = track + trackBase[item] * WV_ROW;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long) gridDim.x * blockDim.x) {
        const long long x = i / WV_ROW;
        const int jj = (int) (i - x * WV_ROW);
        const int k = x == 0 ? 4096 : (int) kidx[it.xOff + x - 1];
        const double *r = rows + (long long) k * CP_ROW;
        double v;
        if (jj < 16) v = r[jj];
        else if (jj == 16) v = r[CP_GAPX] + model[T_GAP_OPEN_X];
        else if (jj == 17) v = r[CP_GAPX] + model[T_GAP_EXTEND_X];
        else if (jj == 18) v = r[CP_GAPX] + model[T_GAP_SWITCH_TO_X];
        else v = r[CP_GAPX];
        dst[i] = v;
    }
}
extern "C" int cpecan_wave_launch_track(hipStream_t stream, const DevItem *items, long long nItems,
                                        const double *track, const long long *trackBase,
                                        const unsigned short *kidx, const double *models, void *states, int maxLX) {
    int bx = (int) ((((long long) maxLX + 1) * WV_ROW + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(cpecan_k_wv_track, dim3(bx, (unsigned) nItems), dim3(256), 0, stream, items, nItems,
                       trackBase, kidx, models, (double *) track);
    if (hipMemsetAsync(states, 0, (size_t) nItems * sizeof(WvState), stream) != hipSuccess) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int cpecan_wave_track_row_doubles(void) { return WV_ROW; }
extern "C" int cpecan_wave_state_bytes(void) { return (int) sizeof(WvState); }
/* the shader clock the forward sweeps of the last run saw, in MHz (s_memtime ticks over 100 MHz s_memrealtime ticks,
 * summed over the first alignments of the batch); 0 when nothing ran */
extern "C" int cpecan_wave_shader_clock_mhz(hipStream_t stream, const void *states, long long nItems, double *mhz) {
    const long long n = nItems < 64 ? nItems : 64;
    std::vector<WvState> h((size_t) n);
    if (hipMemcpyAsync(h.data(), states, (size_t) n * sizeof(WvState), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return -1;
    double c = 0, r = 0;
    for (const WvState &s : h) {
        c += (double) s.clkShader;
        r += (double) s.clkRef;
    }
    *mhz = r > 0 ? 100.0 * c / r : 0.0;
    return 0;
}
/* results of the per-alignment states into the batch's count arrays */
extern "C" __global__ void cpecan_k_wv_counts(const WvState *states, long long nItems, long long *nPairs,
                                              long long *nTot, long long *nCells) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nItems) return;
    nPairs[i] = states[i].nPairs;
    nTot[i] = states[i].nTot;
    nCells[i] = states[i].cells;
}
extern "C" int cpecan_wave_launch_counts(hipStream_t stream, const void *states, long long nItems, long long *nPairs,
                                         long long *nTot, long long *nCells) {
    hipLaunchKernelGGL(cpecan_k_wv_counts, dim3((unsigned) ((nItems + 255) / 256)), dim3(256), 0, stream,
                       (const WvState *) states, nItems, nPairs, nTot, nCells);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif

#if WV_L == 4 && defined(WV_VANILLA)
/* the vanilla machine's track: matrix column x scores the k-mer pair sequence_getKmer2 (impl/pairwiseAligner.c:320-325)
 * exposes for sequence index x - 1 -- a pointer to character max(x - 2, 0): the skip bin looks at the k-mers there
 * and one further, the emissions at the one further (columns 0, 1 and 2 all score k-mers 0 and 1, as in the
 * reference).  Row: per table (match, extra event) mu, sd, 1/sd, K, noise mean, 1/mean, lambda,
 * log(lambda) - log(2 pi); then the bin's five log transition probabilities (cpecan_hip.hip: derive_vanilla) */
extern "C" __global__ void cpecan_k_wv_track_vanilla(const DevItem *__restrict__ items, long long nItems,
                                                     const long long *__restrict__ trackBase,
                                                     const unsigned short *__restrict__ kidx,
                                                     const double *__restrict__ models, double *track) {
    const long long item = blockIdx.y;
    if (item >= nItems) return;
    const DevItem it = items[item];
    const double *hdr = models + (long long) it.model * CP_VMODEL_STRIDE;
    const double *rows = hdr + CP_VHDR;
    const long long n = (it.lX + 1) * WV_ROW;
    double *dst = track + trackBase[item] * WV_ROW;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long) gridDim.x * blockDim.x) {
        const long long x = i / WV_ROW;
        const int jj = (int) (i - x * WV_ROW);
        const long long p = x > 2 ? x - 2 : 0;
        const int kPrev = kidx[it.xOff + p], kCur = kidx[it.xOff + p + 1];
        const double *r = rows + (long long) kCur * CP_VROW;
        double v = 0.0;
        if (jj < 16) {
            const double *q = r + 6 * (jj >> 3);
            switch (jj & 7) {
            case 0: v = q[CP_V_MU]; break;
            case 1: v = q[CP_V_SD]; break;
            case 2: v = q[CP_V_SD] == 0.0 ? 0.0 : 1.0 / q[CP_V_SD]; break;
            case 3: v = q[CP_V_K]; break;
            case 4: v = q[CP_V_NMU]; break;
            case 5: v = 1.0 / q[CP_V_NMU]; break;
            case 6: v = q[CP_V_LAMBDA]; break;
            default: v = q[CP_V_LLAMBDA] - 1.8378770664093453; break;
            }
        } else {
            const double d = fabs(r[CP_V_MU] - rows[(long long) kPrev * CP_VROW + CP_V_MU]);
            long long bin = (long long) (d / 0.5);
            if (bin >= 30) bin = 29;
            v = jj < 21 ? hdr[CP_VHDR_BINS + bin * 5 + (jj - 16)] : (double) bin; /* (entry 21: the bin itself, E-step) */
        }
        dst[i] = v;
    }
}
extern "C" int cpecan_wave_launch_track_vanilla(hipStream_t stream, const DevItem *items, long long nItems,
                                                const double *track, const long long *trackBase,
                                                const unsigned short *kidx, const double *models, void *states,
                                                int maxLX) {
    int bx = (int) ((((long long) maxLX + 1) * WV_ROW + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(cpecan_k_wv_track_vanilla, dim3(bx, (unsigned) nItems), dim3(256), 0, stream, items, nItems,
                       trackBase, kidx, models, (double *) track);
    if (hipMemsetAsync(states, 0, (size_t) nItems * sizeof(WvState), stream) != hipSuccess) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int cpecan_wave_track_row_doubles_vanilla(void) { return WV_ROW; }
#endif

#if WV_L == 4 && defined(WV_HDP)
/* the HDP machine's track: column x (0..lX) = the offset (in doubles) of the table row of the k-mer that matrix
 * column x scores -- sequence_getKmer3 (:327-331): column 0 (index -1) reads the first k-mer, like column 1 -- and
 * the flat gap-X emission log(0.1) (stateMachine.c:1347) plus each of the three transitions into gap X */
extern "C" __global__ void cpecan_k_wv_track_hdp(const DevItem *__restrict__ items, long long nItems,
                                                 const long long *__restrict__ trackBase,
                                                 const int *__restrict__ kid, const DevHdpModel *__restrict__ models,
                                                 double *track) {
    const long long item = blockIdx.y;
    if (item >= nItems) return;
    const DevItem it = items[item];
    const DevHdpModel &m = models[it.model];
    const long long n = (it.lX + 1) * WV_ROW;
    double *dst = track + trackBase[item] * WV_ROW;
    const double px = -2.3025850929940455;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long) gridDim.x * blockDim.x) {
        const long long x = i / WV_ROW;
        const int jj = (int) (i - x * WV_ROW);
        double v = 0.0;
        if (jj == 0) {
            const int id = kid[it.xOff + (x > 0 ? x - 1 : 0)];
            v = id < 0 ? -1.0 : (double) ((long long) m.kmerRow[id] * m.gridLength);
        } else if (jj == 16) v = px + m.t[T_GAP_OPEN_X];
        else if (jj == 17) v = px + m.t[T_GAP_EXTEND_X];
        else if (jj == 18) v = px + m.t[T_GAP_SWITCH_TO_X];
        else if (jj == 19) v = px;
        dst[i] = v;
    }
}
extern "C" int cpecan_wave_launch_track_hdp(hipStream_t stream, const DevItem *items, long long nItems,
                                            const double *track, const long long *trackBase, const int *kid,
                                            const void *models, void *states, int maxLX) {
    int bx = (int) ((((long long) maxLX + 1) * WV_ROW + 255) / 256);
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(cpecan_k_wv_track_hdp, dim3(bx, (unsigned) nItems), dim3(256), 0, stream, items, nItems,
                       trackBase, kid, (const DevHdpModel *) models, (double *) track);
    if (hipMemsetAsync(states, 0, (size_t) nItems * sizeof(WvState), stream) != hipSuccess) return -1;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif

extern "C" int WV_SYM(cpecan_wave_max_width)(void) { return WV_P - 8; }
extern "C" int WV_SYM(cpecan_wave_rows)(void) { return WV_L; }
extern "C" int WV_SYM(cpecan_wave_ring_row_doubles)(void) { return WV_ROW_DOUBLES; }
extern "C" int WV_SYM(cpecan_wave_bring_row_doubles)(void) { return WV_L * 3 * 64; }
/* HBM scratch per alignment: [hit offsets | window totals | their terms | the parked operands | hit masks | candidate list] */
extern "C" long long WV_SYM(cpecan_wave_scratch_bytes)(int ringD) {
    return 2ll * ringD * sizeof(int) + ((long long) ringD / 10 + 8) * (sizeof(WinTotal) + 7 * WV_P * sizeof(double))
           + 4ll * ringD * sizeof(unsigned long long)
           + (long long) WV_L * WV_CAND_PER_DIAG * ringD * (sizeof(int2) + sizeof(double));
}
extern "C" int WV_SYM(cpecan_wave_launch_forward)(hipStream_t stream, const DevItem *items, long long nItems,
                                                  DevParams P, const void *bandTab, const double *track,
                                                  const long long *trackBase, const double *events,
                                                  const double *models, double *Fring, long long ringDoubles,
                                                  int ringD, void *states, int window, int withSwitch) {
    if (withSwitch)
        hipLaunchKernelGGL(WV_SYM(cpecan_k_wv_forward_sw), dim3((unsigned) ((nItems + WV_WPB - 1) / WV_WPB)), dim3(64 * WV_WPB), 0, stream, items, nItems,
                           P, (const int2 *) bandTab, track, trackBase, events, models, Fring, ringDoubles, ringD,
                           (WvState *) states, window);
    else
        hipLaunchKernelGGL(WV_SYM(cpecan_k_wv_forward), dim3((unsigned) ((nItems + WV_WPB - 1) / WV_WPB)), dim3(64 * WV_WPB), 0, stream, items, nItems, P,
                           (const int2 *) bandTab, track, trackBase, events, models, Fring, ringDoubles, ringD,
                           (WvState *) states, window);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int WV_SYM(cpecan_wave_launch_backward)(hipStream_t stream, const DevItem *items, long long nItems,
                                                   DevParams P, const void *bandTab, const double *track,
                                                   const long long *trackBase, const double *models, double *Fring,
                                                   long long ringDoubles, int ringD, void *states, long long *pairs,
                                                   double *pairLogp, long long *totXay, double *totVal, char *scratch,
                                                   long long scratchBytes, double *Bring, int window, int withSwitch) {
#define WV_LAUNCH_B(k)                                                                                            \
    hipLaunchKernelGGL(WV_SYM(k), dim3((unsigned) ((nItems + WV_WPB - 1) / WV_WPB)), dim3(64 * WV_WPB), 0, stream, items, nItems, P, \
                       (const int2 *) bandTab, track, trackBase, models, Fring, ringDoubles, ringD,               \
                       (WvState *) states, pairs, pairLogp, scratch, scratchBytes, Bring, window)
#define WV_LAUNCH_POST                                                                                            \
    hipLaunchKernelGGL(WV_SYM(cpecan_k_wv_post), dim3((unsigned) nItems), dim3(256), 0, stream, items, nItems, P, \
                       (const int2 *) bandTab, models, (const double *) Fring, ringDoubles, ringD,                \
                       (WvState *) states, pairs, pairLogp, totXay, totVal, scratch, scratchBytes, window, 0)
    if (P.mode != 0) {
#if defined(WV_VANILLA)
        if (withSwitch) return -1;
        WV_LAUNCH_B(cpecan_k_wv_backward_em);
        WV_LAUNCH_POST;
#else
        if (withSwitch) WV_LAUNCH_B(cpecan_k_wv_backward_em_sw);
        else WV_LAUNCH_B(cpecan_k_wv_backward_em);
        WV_LAUNCH_POST;
#endif
    } else {
        /* the sweep with decode candidates, the window's totals and decode, then the kernel that sweeps once more
         * the windows whose candidates could not be trusted (it returns at once for the others) */
        if (withSwitch) WV_LAUNCH_B(cpecan_k_wv_backward_sw);
        else WV_LAUNCH_B(cpecan_k_wv_backward);
        WV_LAUNCH_POST;
        if (withSwitch) WV_LAUNCH_B(cpecan_k_wv_resweep_sw);
        else WV_LAUNCH_B(cpecan_k_wv_resweep);
    }
#undef WV_LAUNCH_B
#undef WV_LAUNCH_POST
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#if !defined(WV_VANILLA) && !defined(WV_HDP)
/* what follows the assembly sweep back of a window (cpecan_asm.h), which leaves the refreshes' terms and the decode
 * candidates in scratch as the compiled sweep does: the totals, the decode; then the re-sweep kernel for the windows
 * whose candidates could not be trusted */
extern "C" int WV_SYM(cpecan_wave_launch_post_asm)(hipStream_t stream, const DevItem *items, long long nItems, DevParams P,
                                                   const void *bandTab, const double *track, const long long *trackBase,
                                                   const double *models, double *Fring, long long ringDoubles, int ringD,
                                                   void *states, long long *pairs, double *pairLogp, long long *totXay,
                                                   double *totVal, char *scratch, long long scratchBytes, int window) {
    hipLaunchKernelGGL(WV_SYM(cpecan_k_wv_post), dim3((unsigned) nItems), dim3(256), 0, stream, items, nItems, P,
                       (const int2 *) bandTab, models, (const double *) Fring, ringDoubles, ringD, (WvState *) states, pairs,
                       pairLogp, totXay, totVal, scratch, scratchBytes, window, 0);
    hipLaunchKernelGGL(WV_SYM(cpecan_k_wv_resweep), dim3((unsigned) ((nItems + WV_WPB - 1) / WV_WPB)), dim3(64 * WV_WPB), 0, stream,
                       items, nItems, P, (const int2 *) bandTab, track, trackBase, models, Fring, ringDoubles, ringD,
                       (WvState *) states, pairs, pairLogp, scratch, scratchBytes, (double *) nullptr, window);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif

extern "C" int WV_SYM(cpecan_wave_launch_expect)(hipStream_t stream, const DevItem *items, long long nItems,
                                                 DevParams P, const void *bandTab, const double *track,
                                                 const long long *trackBase, const unsigned short *kidx,
                                                 const double *models, const double *Fring, long long ringDoubles,
                                                 const double *Bring, int ringD, void *states, const char *scratch,
                                                 long long scratchBytes, double *expect, int window, long long *pairs,
                                                 double *pairLogp) {
    hipLaunchKernelGGL(WV_SYM(cpecan_k_wv_expect), dim3((unsigned) nItems, WV_EXPECT_CHUNKS), dim3(WV_P), 0, stream,
                       items, nItems, P, (const int2 *) bandTab, track, trackBase, kidx, models, Fring, ringDoubles,
                       Bring, ringD, (WvState *) states, scratch, scratchBytes, expect, window, pairs, pairLogp);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

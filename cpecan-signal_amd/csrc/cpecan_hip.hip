/*
 * cpecan_hip.hip -- the C-ABI of include/cpecan_hip.h: contexts, model upload, batches.
 *
 * Host side of the thin layer between the reference-shaped C host code and the gfx950 kernels.
 * Nothing here computes DP cells: when no GPU is usable every compute entry point fails with
 * CPECAN_ENODEVICE (there is deliberately no CPU fallback).
 */
#include "cpecan_hip.h"

#include "cpecan_device.h"
#include "cpecan_asm.h"
#include "cpecan_sweep.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <sched.h>
#include <thread>
#include <vector>

extern "C" __global__ void cpecan_k_general(const DevItem *, DevParams, const int *, const int *,
                                            const long long *, const unsigned short *,
                                            const double *, const double *, double *, double *,
                                            long long *, double *, long long *, long long *,
                                            double *, long long *, double *, double *);
extern "C" __global__ void cpecan_k_general5(const DevItem *, DevParams, const int *, const int *,
                                             const long long *, const char *, const char *, const double *,
                                             double *, double *, long long *, double *, long long *,
                                             long long *, double *, long long *, double *, double *);
extern "C" __global__ void cpecan_k_generalv(const DevItem *, DevParams, const int *, const int *,
                                             const long long *, const unsigned short *, const double *,
                                             const double *, const double *, double *, double *,
                                             long long *, double *, long long *, long long *, double *,
                                             long long *, double *);
extern "C" __global__ void cpecan_k_general4(const DevItem *, DevParams, const int *, const int *, const long long *,
                                             const unsigned short *, const double *, const double *, double *, double *,
                                             long long *, double *, long long *, long long *, double *, long long *);
extern "C" __global__ void cpecan_k_generalh(const DevItem *, DevParams, const int *, const int *,
                                             const long long *, const int *, const double *,
                                             const DevHdpModel *, double *, double *, long long *, double *,
                                             long long *, long long *, double *, long long *, double *);
#define W5_DECLARE(L)                                                                                             \
    extern "C" __global__ void cpecan_k_wave5_l##L(const DevItem *, DevParams, const int *, const int *,          \
                                                   const long long *, const char *, const char *, const double *, \
                                                   double *, long long *, double *, long long *, long long *,     \
                                                   double *, long long *, double *);                              \
    extern "C" __global__ void cpecan_k_wave5e_l##L(const DevItem *, DevParams, const int *, const int *,         \
                                                    const long long *, const char *, const char *, const double *, \
                                                    double *, long long *, double *, long long *, long long *,    \
                                                    double *, long long *, double *);                             \
    extern "C" __global__ void cpecan_k_wave5p_l##L(const DevItem *, DevParams, const int *, const int *,         \
                                                    const long long *, const char *, const char *, const double *, \
                                                    double *, long long *, double *, long long *, long long *,    \
                                                    double *, long long *, double *);                             \
    extern "C" __global__ void cpecan_k_wave5pe_l##L(const DevItem *, DevParams, const int *, const int *,        \
                                                     const long long *, const char *, const char *, const double *, \
                                                     double *, long long *, double *, long long *, long long *,   \
                                                     double *, long long *, double *);
W5_DECLARE(1)
W5_DECLARE(2)
W5_DECLARE(3)
extern "C" __global__ void cpecan_k_hdp_kmer_id(const char *, long long, unsigned long long,
                                                unsigned long long, int, int *);
extern "C" __global__ void cpecan_k_kmer_index(const char *, long long, unsigned short *);

extern "C" int cpecan_systolic_max_width(void);
extern "C" int cpecan_systolic_rows(void);
extern "C" int cpecan_systolic_ring_row_doubles(void);
extern "C" int cpecan_systolic_divtest(hipStream_t stream, long long n, unsigned long long seed,
                                       unsigned long long *bad);
extern "C" int cpecan_systolic_state_bytes(void);
extern "C" long long cpecan_systolic_scratch_bytes(int ringD);
extern "C" int cpecan_systolic_launch_track(hipStream_t stream, const DevItem *items, long long nItems,
                                            const double *track, const long long *trackBase,
                                            const unsigned short *kidx, const double *models,
                                            void *states, int maxLX);
extern "C" int cpecan_systolic_launch_forward(hipStream_t stream, const DevItem *items, long long nItems,
                                              DevParams P, const void *bandTab, const double *track,
                                              const long long *trackBase, const double *events,
                                              const double *models, double *Fring,
                                              long long ringDoubles, int ringD, void *states);
extern "C" int cpecan_systolic_launch_backward(hipStream_t stream, const DevItem *items, long long nItems,
                                               DevParams P, const void *bandTab, const double *track,
                                               const long long *trackBase, const double *models,
                                               double *Fring, long long ringDoubles, int ringD,
                                               void *states, long long *pairs, double *pairLogp,
                                               long long *totXay, double *totVal, char *scratch,
                                               long long scratchBytes, double *Bring, int window);
extern "C" int cpecan_systolic_launch_expect(hipStream_t stream, const DevItem *items, long long nItems,
                                             DevParams P, const void *bandTab, const double *track,
                                             const long long *trackBase, const unsigned short *kidx,
                                             const double *models, const double *Fring,
                                             long long ringDoubles, const double *Bring, int ringD,
                                             void *states, const char *scratch, long long scratchBytes,
                                             double *expect, int window, long long *pairs, double *pairLogp);
extern "C" int cpecan_systolic_bring_row_doubles(void);
extern "C" int cpecan_systolic_launch_counts(hipStream_t stream, const void *states, long long nItems,
                                             long long *nPairs, long long *nTot, long long *nCells);
/* the same kernels built with fewer waves per workgroup (symbols suffixed _r1.._r3): bands up to 56, 120, 184
 * k-mers; the narrower the band, the more alignments are resident per CU */
#define SY_DECLARE(sfx)                                                                                         \
    extern "C" int cpecan_systolic_max_width##sfx(void);                                                          \
    extern "C" int cpecan_systolic_ring_row_doubles##sfx(void);                                                   \
    extern "C" int cpecan_systolic_bring_row_doubles##sfx(void);                                                  \
    extern "C" long long cpecan_systolic_scratch_bytes##sfx(int ringD);                                           \
    extern "C" int cpecan_systolic_launch_forward##sfx(hipStream_t, const DevItem *, long long, DevParams,        \
                                                       const void *, const double *, const long long *,           \
                                                       const double *, const double *, double *, long long, int, \
                                                       void *);                                                   \
    extern "C" int cpecan_systolic_launch_backward##sfx(hipStream_t, const DevItem *, long long, DevParams,       \
                                                        const void *, const double *, const long long *,          \
                                                        const double *, double *, long long, int, void *,         \
                                                        long long *, double *, long long *, double *, char *,     \
                                                        long long, double *, int);                                \
    extern "C" int cpecan_systolic_launch_expect##sfx(hipStream_t, const DevItem *, long long, DevParams,         \
                                                      const void *, const double *, const long long *,            \
                                                      const unsigned short *, const double *, const double *,     \
                                                      long long, const double *, int, void *, const char *,       \
                                                      long long, double *, int, long long *, double *);
SY_DECLARE(_r1)
SY_DECLARE(_r2)
SY_DECLARE(_r3)
/* the wave-per-alignment kernels (cpecan_kernel_wave.hip), built for 1..4 cells per lane (symbols _l1.._l4) */
#define WV_DECLARE(sfx)                                                                                         \
    extern "C" int cpecan_wave_max_width##sfx(void);                                                              \
    extern "C" int cpecan_wave_ring_row_doubles##sfx(void);                                                       \
    extern "C" int cpecan_wave_bring_row_doubles##sfx(void);                                                      \
    extern "C" long long cpecan_wave_scratch_bytes##sfx(int ringD);                                               \
    extern "C" int cpecan_wave_launch_forward##sfx(hipStream_t, const DevItem *, long long, DevParams,            \
                                                   const void *, const double *, const long long *,               \
                                                   const double *, const double *, double *, long long, int,     \
                                                   void *, int, int);                                             \
    extern "C" int cpecan_wave_launch_backward##sfx(hipStream_t, const DevItem *, long long, DevParams,           \
                                                    const void *, const double *, const long long *,              \
                                                    const double *, double *, long long, int, void *,             \
                                                    long long *, double *, long long *, double *, char *,         \
                                                    long long, double *, int, int);                               \
    extern "C" int cpecan_wave_launch_expect##sfx(hipStream_t, const DevItem *, long long, DevParams,             \
                                                  const void *, const double *, const long long *,                \
                                                  const unsigned short *, const double *, const double *,         \
                                                  long long, const double *, int, void *, const char *,           \
                                                  long long, double *, int, long long *, double *);
WV_DECLARE(_l2)
WV_DECLARE(_l3)
WV_DECLARE(_l4)
/* ... and the same sweeps for the HDP signal machine (-DWV_HDP, symbols _h2.._h4) */
WV_DECLARE(_h2)
WV_DECLARE(_h3)
WV_DECLARE(_h4)
/* ... and for the vanilla signal machine (-DWV_VANILLA, symbols _v2, _v3; its four-cell build is built and linked,
 * cpecan_kernel_wave_v4.o, but not used: it does not fit the register file without spilling, and bands above 184
 * k-mers run on the general kernel) */
WV_DECLARE(_v2)
WV_DECLARE(_v3)
extern "C" int cpecan_wave_launch_track_vanilla(hipStream_t stream, const DevItem *items, long long nItems,
                                                const double *track, const long long *trackBase,
                                                const unsigned short *kidx, const double *models, void *states,
                                                int maxLX);
extern "C" int cpecan_wave_track_row_doubles_vanilla(void);
extern "C" int cpecan_wave_launch_track_hdp(hipStream_t stream, const DevItem *items, long long nItems,
                                            const double *track, const long long *trackBase, const int *kid,
                                            const void *models, void *states, int maxLX);
extern "C" int cpecan_wave_launch_track(hipStream_t stream, const DevItem *items, long long nItems,
                                        const double *track, const long long *trackBase,
                                        const unsigned short *kidx, const double *models, void *states, int maxLX);
extern "C" int cpecan_wave_track_row_doubles(void);
extern "C" int cpecan_wave_state_bytes(void);
extern "C" int cpecan_wave_shader_clock_mhz(hipStream_t stream, const void *states, long long nItems, double *mhz);
extern "C" int cpecan_wave_launch_counts(hipStream_t stream, const void *states, long long nItems, long long *nPairs,
                                         long long *nTot, long long *nCells);

extern "C" int cpecan_wave_launch_post_asm_l3(hipStream_t stream, const DevItem *items, long long nItems, DevParams P,
                                              const void *bandTab, const double *track, const long long *trackBase,
                                              const double *models, double *Fring, long long ringDoubles, int ringD,
                                              void *states, long long *pairs, double *pairLogp, long long *totXay,
                                              double *totVal, char *scratch, long long scratchBytes, int window);

struct SyBuild { /* one build of the throughput kernels */
    int rows;  /* waves per workgroup (systolic) or cells per lane (wave) */
    bool wave; /* one wave per alignment (cpecan_kernel_wave.hip) */
    int (*max_width)(void);
    int (*ring_row_doubles)(void);
    int (*bring_row_doubles)(void);
    long long (*scratch_bytes)(int);
    int (*launch_forward)(hipStream_t, const DevItem *, long long, DevParams, const void *, const double *,
                          const long long *, const double *, const double *, double *, long long, int, void *, int, int);
    int (*launch_backward)(hipStream_t, const DevItem *, long long, DevParams, const void *, const double *,
                           const long long *, const double *, double *, long long, int, void *, long long *, double *,
                           long long *, double *, char *, long long, double *, int, int);
    decltype(&cpecan_systolic_launch_expect) launch_expect;
};
#define SY_BUILD(r, sfx)                                                                                          \
    { r, false, cpecan_systolic_max_width##sfx, cpecan_systolic_ring_row_doubles##sfx,                            \
      cpecan_systolic_bring_row_doubles##sfx, cpecan_systolic_scratch_bytes##sfx,                                 \
      [](hipStream_t st, const DevItem *it, long long n, DevParams P, const void *bt, const double *tr,           \
         const long long *tb, const double *ev, const double *mo, double *F, long long rd, int D, void *S, int,   \
         int) {                                                                                                   \
          return cpecan_systolic_launch_forward##sfx(st, it, n, P, bt, tr, tb, ev, mo, F, rd, D, S);              \
      },                                                                                                          \
      [](hipStream_t st, const DevItem *it, long long n, DevParams P, const void *bt, const double *tr,           \
         const long long *tb, const double *mo, double *F, long long rd, int D, void *S, long long *pa,           \
         double *pl, long long *tx, double *tv, char *sc, long long sb, double *B, int w, int) {                  \
          return cpecan_systolic_launch_backward##sfx(st, it, n, P, bt, tr, tb, mo, F, rd, D, S, pa, pl, tx, tv,  \
                                                      sc, sb, B, w);                                              \
      },                                                                                                          \
      cpecan_systolic_launch_expect##sfx }
#define WV_BUILD(r, sfx)                                                                                          \
    { r, true, cpecan_wave_max_width##sfx, cpecan_wave_ring_row_doubles##sfx, cpecan_wave_bring_row_doubles##sfx, \
      cpecan_wave_scratch_bytes##sfx, cpecan_wave_launch_forward##sfx, cpecan_wave_launch_backward##sfx,          \
      cpecan_wave_launch_expect##sfx }
static const SyBuild SY_BUILDS[4] = { SY_BUILD(1, _r1), SY_BUILD(2, _r2), SY_BUILD(3, _r3), SY_BUILD(4, ) };
/* (a one-cell-per-lane build would only serve bands below 57 k-mers; the two-cell build takes those too) */
static const SyBuild WV_BUILDS[4] = { WV_BUILD(2, _l2), WV_BUILD(2, _l2), WV_BUILD(3, _l3), WV_BUILD(4, _l4) };
static const SyBuild HV_BUILDS[4] = { WV_BUILD(2, _h2), WV_BUILD(2, _h2), WV_BUILD(3, _h3), WV_BUILD(4, _h4) };
static const SyBuild VV_BUILDS[4] = { WV_BUILD(2, _v2), WV_BUILD(2, _v2), WV_BUILD(3, _v3), WV_BUILD(3, _v3) };
/* which family a batch runs on: the wave kernels unless CPECAN_KERNELS=systolic asks for the workgroup-per-alignment ones */
static bool use_wave_kernels() {
    const char *k = getenv("CPECAN_KERNELS");
    return !(k && strcmp(k, "systolic") == 0);
}


namespace {

thread_local std::string g_err;

/* std::vector without the zero-fill of resize(): the big host tables here are written in full right after they are
 * sized (by several threads, so the page faults spread too) */
template <class T> struct NoInit : std::allocator<T> {
    template <class U> struct rebind { using other = NoInit<U>; };
    template <class U, class... A> void construct(U *p, A &&...a) {
        if constexpr (sizeof...(A) == 0) ::new ((void *) p) U;
        else ::new ((void *) p) U(std::forward<A>(a)...);
    }
};

/* CPECAN_TIMING=1: wall-clock laps of the host-side set-up calls on stderr (where the time before the first kernel goes) */
struct Lap {
    const char *who;
    bool on;
    std::chrono::steady_clock::time_point t, t0;
    explicit Lap(const char *w) : who(w), on(getenv("CPECAN_TIMING") != nullptr), t(std::chrono::steady_clock::now()), t0(t) {}
    ~Lap() {
        if (on)
            fprintf(stderr, "[cpecan timing] %s: TOTAL %.1f ms\n", who,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    void operator()(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[cpecan timing] %s: %s %.1f ms\n", who, what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

/* worker threads for host-side table derivation: the CPUs this process may run on (its affinity mask, which is
 * what a job's CPU share shows up as), at most 32 -- a node's hardware_concurrency() is the whole machine, and
 * eight ranks of a multi-GPU job each spawning that many threads would run into the host's task limits */
int host_threads() {
    int n = (int) std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0 && c < n) n = c;
    }
    return n < 1 ? 1 : n > 32 ? 32 : n;
}

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(CPECAN_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                        __FILE__, __LINE__);                                                 \
    } while (0)

/* one candidate pair on its way to the host: its coordinates.  With it goes the device's verdict (an int, see
 * cpecan_k_pack_pairs); the exponent (F + B) - totalProbability stays in HBM and is fetched for the few candidates the
 * host has to settle itself, and for callers that ask for it: 12 bytes per candidate cross PCIe instead of 20. */
struct PackedPair {
    int x, y;
};
#define CP_UNDECIDED_CAP 65536ull /* candidates per batch the host settles with its libm before it fetches exponents item by item */

/* Device memory of batches and model tables goes through a small caching allocator: hipMalloc and hipFree wait for
 * the device, so a host thread that prepares the next batch while the GPU works on the current one (one-shot
 * alignment of a stream of batches) would otherwise stall on every buffer.  A released block is kept (up to
 * CPECAN_ALLOC_CACHE_GB; by default half of the device's memory -- one process per GPU is the deployment, and RCCL's
 * buffers, torch in the same process or other processes on the card keep the other half; cpecan_hip_trim_cache()
 * gives everything back)
 * and handed to the next request it fits within 25 %.  A block keeps its real size through every reuse. */
struct DevCache {
    struct Block { void *p; size_t bytes; int device; };
    std::mutex lock;
    std::vector<Block> blocks;
    size_t held = 0;
    const bool pinnedHost; /* the same for pinned host memory (the packed pairs of a batch): pinning and unpinning
                              150 MB per batch costs tens of milliseconds; up to CPECAN_PINNED_CACHE_GB, default 8 */
    size_t capBytes = 0; /* 0: not worked out yet */
    explicit DevCache(bool host) : pinnedHost(host) {}
    size_t cap() { /* (under `lock`) */
        if (capBytes == 0) {
            const char *e = getenv(pinnedHost ? "CPECAN_PINNED_CACHE_GB" : "CPECAN_ALLOC_CACHE_GB");
            double gb = e ? atof(e) : 8.0;
            if (!e && !pinnedHost) {
                size_t freeB = 0, totalB = 0;
                /* half the card: three C3 batches on the assembly sweeps (ring of three windows: 45 GB each) alive at once,
                 * as a one-shot service keeps them, still turn over inside the cache */
                gb = hipMemGetInfo(&freeB, &totalB) == hipSuccess ? (double) totalB / 2.0 / (double) (1ull << 30) : 32.0;
            }
            capBytes = (size_t) (gb * (double) (1ull << 30)) + 1;
        }
        return capBytes;
    }
    hipError_t raw_alloc(void **out, size_t bytes) { return pinnedHost ? hipHostMalloc(out, bytes, hipHostMallocDefault) : hipMalloc(out, bytes); }
    void raw_free(void *p) { (void) (pinnedHost ? hipHostFree(p) : hipFree(p)); }
    hipError_t get(void **out, size_t bytes, size_t *got) {
        *got = bytes;
        int device = 0;
        (void) hipGetDevice(&device);
        {
            std::lock_guard<std::mutex> g(lock);
            size_t best = blocks.size();
            for (size_t i = 0; i < blocks.size(); i++)
                if (blocks[i].device == device && blocks[i].bytes >= bytes && blocks[i].bytes <= bytes + bytes / 4 + 4096 &&
                    (best == blocks.size() || blocks[i].bytes < blocks[best].bytes))
                    best = i;
            if (best != blocks.size()) {
                *out = blocks[best].p;
                *got = blocks[best].bytes;
                held -= blocks[best].bytes;
                blocks.erase(blocks.begin() + (long) best);
                return hipSuccess;
            }
        }
        hipError_t e = raw_alloc(out, bytes);
        if (e != hipSuccess) { /* out of memory with blocks in the cache: give them back and try once more */
            trim(0);
            (void) hipGetLastError();
            e = raw_alloc(out, bytes);
        }
        return e;
    }
    void put(void *p, size_t bytes) {
        int device = 0;
        (void) hipGetDevice(&device);
        std::lock_guard<std::mutex> g(lock);
        if (held + bytes > cap()) { /* the cache is bounded (small blocks are kept too: hipFree waits for the device
                                       whatever the size, and the device is busy with the previous batch) */
            raw_free(p);
            return;
        }
        blocks.push_back({ p, bytes, device });
        held += bytes;
    }
    void trim(size_t keep) {
        std::lock_guard<std::mutex> g(lock);
        while (!blocks.empty() && held > keep) {
            raw_free(blocks.back().p);
            held -= blocks.back().bytes;
            blocks.pop_back();
        }
    }
};
DevCache &dev_cache() {
    static DevCache *c = new DevCache(false); /* (never destroyed: the runtime may be gone by the time statics are) */
    return *c;
}
DevCache &pinned_cache() {
    static DevCache *c = new DevCache(true);
    return *c;
}

/* a block of pinned host memory from the cache (host-built tables on their way to the device) */
template <typename T> struct PinnedBuf {
    T *p = nullptr;
    size_t n = 0, blockBytes = 0;
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return pinned_cache().get((void **) &p, count * sizeof(T), &blockBytes);
    }
    void release() {
        if (p) pinned_cache().put(p, blockBytes);
        p = nullptr;
        n = blockBytes = 0;
    }
    ~PinnedBuf() { release(); }
};

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0, blockBytes = 0;
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return dev_cache().get((void **) &p, count * sizeof(T), &blockBytes);
    }
    void release() {
        if (p) dev_cache().put(p, blockBytes);
        p = nullptr;
        n = blockBytes = 0;
    }
    void swap(DevBuf &o) {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(blockBytes, o.blockBytes);
    }
    ~DevBuf() { release(); }
};

/* Declared after a function's own DevBuf / PinnedBuf objects and before its first asynchronous use of them: whichever
 * way the function returns, the streams it fed are idle before those buffers go back to the cache (a released block
 * can be handed to another thread at once; hipFree used to wait for the device here). */
struct StreamFence {
    hipStream_t a = nullptr, b = nullptr;
    ~StreamFence() {
        if (a) (void) hipStreamSynchronize(a);
        if (b) (void) hipStreamSynchronize(b);
    }
};

} // namespace

#define CP_WAVE5_PAIRED_BELOW 1536 /* alignments: below this (fewer than 1.5 per SIMD) the 5-state machine runs on two waves
                                    * per alignment: 1.14-1.24x at 1024 alignments, 0.8-0.9x at 4096 */
struct cpecan_ctx {
    int device = 0;
    long long modelEpoch = 0; /* counts cpecan_hip_models_clear calls */
    hipStream_t stream = nullptr;
    /* input preparation (uploads, table assembly, k-mer indices) goes through a stream of the highest priority: it
     * gets a hardware queue of its own and its copies and small kernels are not held up behind the sweeps of the
     * batches that are running while the next one is prepared; every call that uses it waits for it before it returns */
    hipStream_t prep = nullptr;
    DevBuf<double> models; /* nModels * CP_MODEL_STRIDE */
    void *pinned = nullptr; /* staging slots of cpecan_hip_models_create */
    size_t pinnedBytes = 0;
    std::vector<double> switchToX; /* per strawMan model: its GAP_SWITCH_TO_X (the tables themselves live on the device only) */
    int nModels = 0;
    DevBuf<double> models5; /* 5-state symbol models, nModels5 * CP_MODEL5_STRIDE */
    std::vector<double> hostModels5;
    int nModels5 = 0;
    /* HDP models: descriptors on the device, their tables in buffers of their own */
    struct HdpTables {
        DevBuf<int> kmerRow;
        DevBuf<double> grid, y, slope;
    };
    std::vector<HdpTables *> hdpTables;
    std::vector<DevHdpModel> hostModelsH;
    DevBuf<DevHdpModel> modelsH;
    std::string hdpAlphabet;
    DevBuf<double> modelsV; /* vanilla signal models, nModelsV * CP_VMODEL_STRIDE */
    DevBuf<double> models4; /* 4-state signal models: strawMan tables whose header holds eleven transitions */
    std::vector<double> hostModels4;
    int nModels4 = 0;
    std::vector<double> hostModelsV;
    int nModelsV = 0;
};

struct cpecan_batch {
    cpecan_ctx *ctx = nullptr;
    int64_t nItems = 0;
    int mode = 0, kernel = 0, flags = 0;
    DevParams P{};
    std::vector<DevItem> hItems;
    DevBuf<DevItem> items;
    DevBuf<int> bandL, bandR;
    DevBuf<long long> cellPrefix;
    DevBuf<char> chars, charsY; /* charsY: DNA batches (5-state machine) */
    bool dna = false;
    bool vanilla = false, hdp = false, sm4 = false;
    DevBuf<double> logNoise; /* vanilla batches: log(event noise), host libm */
    DevBuf<int> kid;         /* HDP batches: k-mer id over the model's alphabet per X position */
    DevBuf<unsigned short> kidx;
    DevBuf<double> events;
    DevBuf<double> Fstore, Bstore, dbgB;
    DevBuf<double> Bring; /* systolic Baum-Welch: backward cells of one window per item */
    DevBuf<long long> pairs;
    DevBuf<double> pairLogp;
    DevBuf<long long> nPairs, totXay, nTot, nCells;
    DevBuf<double> totVal;
    DevBuf<double> expect;
    DevBuf<int> workCounter;
    DevBuf<char> syStates, syScratch;
    DevBuf<int> bandTab; /* systolic kernels: (first, last) matrix column of every diagonal of every item */
    long long scratchBytes = 0;
    int nWindows = 0;
    DevBuf<double> track;
    DevBuf<long long> trackBase;
    long long ringDoubles = 0;
    int ringD = 0, maxLX = 0;
    int nWorkers = 0, maxWidth = 0;
    const SyBuild *sy = &SY_BUILDS[3]; /* systolic path: the build of the kernels the batch runs on */
    int device = 0;                    /* the context's device, kept for the destructor */
    int nModels = 0;
    int expectLen = CPECAN_EXPECTATION_LEN; /* doubles per model in `expect` */
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    /* systolic path: the batch runs as nGroups independent groups of alignments, each on a stream of
     * its own, so that the tail of one group's kernel overlaps the other groups' kernels (a launch
     * lasts as long as its slowest workgroup).  evStage: per group, one event after every kernel. */
    int nGroups = 1;
    std::vector<hipStream_t> gStream, gStreamB; /* gStreamB: the wave kernels' backward sweeps (see batch_run) */
    hipStream_t asmPost = nullptr;     /* assembly sweeps: the totals and the decode of a window, beside the next window's sweeps */
    std::vector<hipEvent_t> evPost;    /* ... done, per window */
    bool gStreamOwned = true;
    long long modelEpoch = 0; /* the context's when the batch was created */
    int stateBytes = 0;
    std::vector<hipEvent_t> evStage, evJoin;
    hipEvent_t evFork = nullptr;
    std::vector<long long> hNPairs, hNTot, hNCells;
    /* aligned pairs as the callers get them: the device selects by the exponent with a margin, the host finishes
     * exp(), the threshold test and floor(p * 1e7) with the reference's libm (impl/pairwiseAligner.c:776-786) */
    std::vector<long long> hPairs; /* triples, packed per item at hPairBase */
    std::vector<double> hLogp;
    std::vector<long long> hPairBase;
    /* the way back to the host: the candidates of all items packed into one device buffer of 16-byte records and
     * copied in one piece into pinned memory */
    DevBuf<long long> packBase;
    DevBuf<PackedPair> packed;
    DevBuf<int> packedPost;
    DevBuf<long long> undecided;   /* cpecan_k_pack_pairs: [count | CP_UNDECIDED_CAP x (packed index, exponent bits)] */
    long long *hUndecided = nullptr; /* its pinned copy */
    PackedPair *hPacked = nullptr; /* hipHostMalloc */
    int *hPost = nullptr;          /* hipHostMalloc: the device's verdict per candidate (cpecan_k_pack_pairs) */
    size_t hPackedCap = 0;
    size_t hPackedBlock = 0, hPostBlock = 0, hUndecidedBlock = 0; /* the real sizes of those blocks (the allocator's cache) */
    int trackRow = CP_ROW; /* doubles per column of the track */
    bool countsValid = false, ran = false;
    bool packedInRun = false; /* the last run ended with cpecan_k_pack_base + cpecan_k_pack_pairs */
    bool compactPairs = false; /* every sequence of the batch is shorter than 65536 elements: a packed candidate crosses
                                  PCIe as (x | y << 16) and its verdict, 8 bytes instead of 12 */
    /* the hand-scheduled assembly sweeps (cpecan_asm.h): the host's plan of windows and band steps, the forward waves'
     * contexts */
    bool useAsm = false, asmBackward = false;
    int asmMaxWindows = 0;
    DevBuf<AsmPlanWin> planWin;
    DevBuf<AsmPlanCtl> planCtl;
    DevBuf<long long> planOff;
    DevBuf<char> asmCtx;
    DevBuf<unsigned> asmMasks;
};

/* The plan of one alignment for the assembly sweeps: its traceback windows (getPosteriorProbsWithBanding's schedule,
 * impl/pairwiseAligner.c:917-921 -- a function of the band alone; the same walk as the window count in batch_create) and,
 * per diagonal, whether the band's edges step and whether the forward sweep keeps all three states of the diagonal
 * (forward_window() of cpecan_kernel_wave.hip works the same rule out on the device).  tab: (first, last) column per
 * diagonal. */
static void build_asm_plan(const int *tab, long long nDiag, const cpecan_band_params &bp, std::vector<AsmPlanWin> &wins,
                           AsmPlanCtl *ctl) {
    const long long D = nDiag - 1;
    memset(ctl, 0, (size_t) (D / ASM_BLOCK + 2) * sizeof(AsmPlanCtl));
    wins.clear();
    long long tracedBackTo = 0, cells = 1;
    int d0 = 0;
    for (long long k = 1; k <= D; k++) {
        const int xmn = tab[2 * k], xmx = tab[2 * k + 1];
        if (xmn != tab[2 * k - 2]) ctl[k >> 6].stepMin |= 1ull << (k & 63);
        if (xmx != tab[2 * k - 1]) ctl[k >> 6].stepMax |= 1ull << (k & 63);
        cells += xmx - xmn + 1;
        const bool atEnd = k == D;
        const bool tb = k >= tracedBackTo + bp.minDiagsBetweenTraceBack && xmx - xmn + 1 <= bp.diagonalExpansion * 2 + 1;
        if (!(atEnd || tb)) continue;
        AsmPlanWin w{};
        w.d0 = d0;
        w.top = (int) k;
        w.from = (int) (k - (atEnd ? 0 : bp.traceBackDiagonals + 1));
        w.to = (int) tracedBackTo;
        w.atEnd = atEnd ? 1 : 0;
        w.xminTop = xmn;
        w.xmaxTop = xmx;
        w.cells = cells;
        w.xmin0 = tab[2 * d0];
        w.xmax0 = tab[2 * d0 + 1];
        w.tpost0 = std::min(w.top, w.from);
        wins.push_back(w);
        d0 = (int) k;
        tracedBackTo = w.from;
    }
    for (size_t wi = 0; wi < wins.size(); wi++) {
        AsmPlanWin &w = wins[wi];
        w.nWindows = (int) wins.size();
        /* all three states: the two diagonals a launch resumes from and every diagonal with a totalProbability refresh of
         * the window that decodes it (every 10th decoded diagonal, counted down from the window's first); the compiled
         * sweep back (CPECAN_ASM=1: tests, timing) also reads the diagonal below a refresh, the assembly one does not
         * (gen_sweeps.py: the refresh's second half is F.match + B.match of the diagonal above); everything where windows
         * are shorter than the traceback margin */
        const bool belowToo = getenv("CPECAN_ASM") != nullptr && atoi(getenv("CPECAN_ASM")) == 1;
        const bool endW = w.atEnd != 0;
        const int tpA = w.tpost0;
        int tpB = tpA;
        bool allFull = false;
        if (!endW) {
            tpB = wins[wi + 1].tpost0;
            allFull = tpB < w.top;
        }
        for (int dj = w.d0 + 1; dj <= w.top; dj++) {
            const int rA = ((tpA - dj) % 10 + 10) % 10, rB = endW ? 99 : ((tpB - dj) % 10 + 10) % 10;
            const int rHere = dj <= w.from ? rA : rB, rAbove = dj + 1 <= w.from ? rA : rB;
            if (allFull || dj >= w.top - 1 || rHere == 0 || (belowToo && rAbove == 1)) ctl[dj >> 6].full |= 1ull << (dj & 63);
        }
    }
}

extern "C" __global__ void cpecan_k_pack_pairs(const DevItem *items, const long long *packBase, const long long *pairs,
                                               const double *logp, double threshold, long long capacity,
                                               PackedPair *out, int *post, long long *undecided, int compact);
extern "C" __global__ void cpecan_k_pack_base(const DevItem *items, const long long *nPairs, long long nItems,
                                              long long *packBase);

extern "C" {

const char *cpecan_hip_last_error(void) { return g_err.c_str(); }
const char *cpecan_hip_version(void) { return "cpecan-signal_amd 0.1 (gfx950)"; }

int cpecan_hip_device_count(int *count) {
    if (!count) return fail(CPECAN_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(CPECAN_ENODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return CPECAN_OK;
}

int cpecan_hip_ctx_create(int device, cpecan_ctx **out) {
    if (!out) return fail(CPECAN_EINVAL, "ctx is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(CPECAN_ENODEVICE, "no HIP device is available; this library has no CPU path");
    if (device < 0 || device >= n) return fail(CPECAN_EINVAL, "device %d out of range (%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    cpecan_ctx *c = new (std::nothrow) cpecan_ctx();
    if (!c) return fail(CPECAN_EINVAL, "out of host memory");
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(CPECAN_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    int least = 0, greatest = 0;
    (void) hipDeviceGetStreamPriorityRange(&least, &greatest);
    e = hipStreamCreateWithPriority(&c->prep, hipStreamNonBlocking, greatest);
    if (e != hipSuccess) {
        (void) hipStreamDestroy(c->stream);
        delete c;
        return fail(CPECAN_EHIP, "hipStreamCreateWithPriority: %s", hipGetErrorString(e));
    }
    *out = c;
    return CPECAN_OK;
}

int cpecan_hip_ctx_destroy(cpecan_ctx *c) {
    if (!c) return CPECAN_OK;
    (void) hipSetDevice(c->device);
    /* the model tables go back to the allocator's cache below: nothing queued through this context may still read them
     * (a batch of this context that is still running on streams of its own is the caller's to finish first) */
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    if (c->prep) (void) hipStreamSynchronize(c->prep);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    if (c->prep) (void) hipStreamDestroy(c->prep);
    if (c->pinned) (void) hipHostFree(c->pinned);
    for (auto *t : c->hdpTables) delete t;
    c->hdpTables.clear();
    delete c;
    (void) hipGetLastError();
    return CPECAN_OK;
}

int cpecan_hip_trim_cache(void) {
    dev_cache().trim(0);
    pinned_cache().trim(0);
    (void) hipGetLastError();
    return CPECAN_OK;
}

int cpecan_hip_ctx_stream(cpecan_ctx *c, void **stream) {
    if (!c || !stream) return fail(CPECAN_EINVAL, "NULL argument");
    *stream = (void *) c->stream;
    return CPECAN_OK;
}

/* One derived row per k-mer.  K = log_inv_sqrt_2pi - log(sigma) is the part of
 * emissions_signal_logGaussPdf (impl/stateMachine.c:333-343) that does not depend on the event;
 * evaluated here with the host libm exactly as the reference's per-cell code would. */
static void derive_rows(const cpecan_sm3_model *m, double *dst) {
    const double c = -0.91893853320467267;
    for (int i = 0; i < 9; i++) dst[i] = m->transitions[i];
    for (int i = 9; i < CP_MODEL_HEADER; i++) dst[i] = 0.0;
    double *rows = dst + CP_MODEL_HEADER;
    for (int k = 0; k <= CPECAN_NUM_KMERS; k++) {
        double *r = rows + (size_t) k * CP_ROW;
        if (k == CPECAN_NUM_KMERS) { /* "not a k-mer": model reads 0.0, gap prob LOG_ZERO (:185,:223) */
            for (int j = 0; j < CP_ROW; j++) r[j] = 0.0;
            r[CP_K1] = r[CP_K2] = r[CP_YK1] = r[CP_YK2] = -INFINITY;
            r[CP_GAPX] = -INFINITY;
            continue;
        }
        const double *a = m->match_probs + 1 + (size_t) k * CPECAN_MODEL_PARAMS;
        const double *b = m->gap_y_probs + 1 + (size_t) k * CPECAN_MODEL_PARAMS;
        const double sd[4] = { a[1], a[3], b[1], b[3] };
        const double mu[4] = { a[0], a[2], b[0], b[2] };
        for (int g = 0; g < 4; g++) {
            double *q = r + 4 * g;
            q[0] = mu[g];
            q[1] = sd[g];
            q[2] = sd[g] == 0.0 ? 0.0 : 1.0 / sd[g];
            q[3] = sd[g] == 0.0 ? -INFINITY : c - log(sd[g]);
        }
        r[CP_GAPX] = m->gap_x_probs[k];
        r[17] = 0.0;
    }
}

/* Room for n more strawMan models at the end of the device table: a new block, the old rows copied across on the
 * device (no host mirror of the tables is kept).  *fresh receives the device address of the first new model. */
static int grow_models(cpecan_ctx *c, int32_t n, double **fresh) {
    const size_t old = (size_t) c->nModels * CP_MODEL_STRIDE, total = old + (size_t) n * CP_MODEL_STRIDE;
    if (c->stream) (void) hipStreamSynchronize(c->stream); /* (the old table goes back to the allocator's cache) */
    DevBuf<double> grown;
    hipError_t e = grown.alloc(total);
    if (e != hipSuccess) return fail(CPECAN_EHIP, "model table allocation: %s", hipGetErrorString(e));
    {
        StreamFence fence{ c->prep, nullptr };
        /* on the stream the uploads that follow use, and over before the old block is released */
        if (old) HIP_TRY(hipMemcpyAsync(grown.p, c->models.p, old * sizeof(double), hipMemcpyDeviceToDevice, c->prep));
    }
    grown.swap(c->models);
    *fresh = c->models.p + old;
    return CPECAN_OK;
}

int cpecan_hip_models_create(cpecan_ctx *c, const cpecan_sm3_model *models, int32_t n,
                             int32_t threads, int32_t *ids) {
    if (!c || !models || n <= 0 || !ids) return fail(CPECAN_EINVAL, "bad argument");
    for (int i = 0; i < n; i++)
        if (!models[i].match_probs || !models[i].gap_x_probs || !models[i].gap_y_probs)
            return fail(CPECAN_EINVAL, "model %d has a NULL table", i);
    HIP_TRY(hipSetDevice(c->device));
    Lap lap("models_create");
    int nt = threads > 0 ? threads : host_threads();
    nt = std::max(1, std::min(nt, (int) n));
    double *fresh = nullptr;
    int rc = grow_models(c, n, &fresh);
    if (rc != CPECAN_OK) return rc;
    /* every host thread derives a model into one of its two pinned slots and sends it on its way; the slot is
     * written again once its copy has gone (no host copy of the whole table exists at any time) */
    const size_t slotBytes = CP_MODEL_STRIDE * sizeof(double), want = slotBytes * 2 * (size_t) nt;
    if (c->pinnedBytes < want) {
        if (c->pinned) (void) hipHostFree(c->pinned);
        c->pinned = nullptr;
        c->pinnedBytes = 0;
        HIP_TRY(hipHostMalloc(&c->pinned, want, hipHostMallocDefault));
        c->pinnedBytes = want;
    }
    std::vector<hipEvent_t> gone(2 * (size_t) nt, nullptr);
    for (auto &ev : gone) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    lap("device table, pinned slots");
    std::atomic<int> bad{0};
    std::vector<std::thread> pool;
    for (int w = 0; w < nt; w++)
        pool.emplace_back([&, w]() {
            if (hipSetDevice(c->device) != hipSuccess) { bad = 1; return; }
            int turn = 0;
            for (int i = w; i < n; i += nt, turn++) {
                const size_t slot = 2 * (size_t) w + (turn & 1);
                double *dst = (double *) ((char *) c->pinned + slot * slotBytes);
                if (turn >= 2 && hipEventSynchronize(gone[slot]) != hipSuccess) { bad = 1; return; }
                derive_rows(&models[i], dst);
                if (hipMemcpyAsync(fresh + (size_t) i * CP_MODEL_STRIDE, dst, slotBytes, hipMemcpyHostToDevice, c->prep) != hipSuccess ||
                    hipEventRecord(gone[slot], c->prep) != hipSuccess) { bad = 1; return; }
            }
        });
    for (auto &t : pool) t.join();
    hipError_t se = hipStreamSynchronize(c->prep);
    for (auto &ev : gone) (void) hipEventDestroy(ev);
    if (bad || se != hipSuccess) {
        c->models.release(); /* the table is in an unknown state: the context's strawMan models are gone */
        c->switchToX.clear();
        c->nModels = 0;
        c->modelEpoch++;
        return fail(CPECAN_EHIP, "model table upload failed: %s", hipGetErrorString(se != hipSuccess ? se : hipGetLastError()));
    }
    lap("derive rows (threads) || upload");
    for (int i = 0; i < n; i++) {
        ids[i] = c->nModels + i;
        c->switchToX.push_back(models[i].transitions[T_GAP_SWITCH_TO_X]);
    }
    c->nModels += n;
    return CPECAN_OK;
}

/* One element of one read's derived table from the base model's derived table, the read's scaling parameters
 * (emissions_signal_scaleModel impl/stateMachine.c:631-651) and the three values per k-mer the host took with its
 * libm (K1, the scaled noise sd, K2): every other entry is one IEEE multiply, add or divide, rounded as on the host. */
extern "C" __global__ void cpecan_k_scale_models(const double *base, const double *scalings /* n x 5 */,
                                                 const double *hostPart /* n x 4096 x 3 */, int n, double *out) {
    const long long e = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= CP_MODEL_STRIDE) return;
    const double b = base[e];
    const long long r = e - CP_MODEL_HEADER;
    const int k = r >= 0 ? (int) (r / CP_ROW) : -1, j = r >= 0 ? (int) (r % CP_ROW) : -1;
    const bool plain = k < 0 || k >= CPECAN_NUM_KMERS || j >= 8;
    /* the neighbours a derived entry needs: the level sd (j 1, 2), the noise mean's row mates */
    const double sd = plain ? 0.0 : base[CP_MODEL_HEADER + (long long) k * CP_ROW + 1];
    for (int m = blockIdx.y; m < n; m += gridDim.y) {
        double v = b;
        if (!plain) {
            const double *sc = scalings + 5 * (long long) m;
            const double *h = hostPart + ((long long) m * CPECAN_NUM_KMERS + k) * 3;
            switch (j) {
            case 0: v = __dadd_rn(__dmul_rn(b, sc[0]), sc[1]); break;
            case 1: v = __dmul_rn(b, sc[2]); break;
            case 2: { const double s = __dmul_rn(sd, sc[2]); v = s == 0.0 ? 0.0 : __ddiv_rn(1.0, s); break; }
            case 3: v = h[0]; break;
            case 4: v = __dmul_rn(b, sc[3]); break;
            case 5: v = h[1]; break;
            case 6: v = h[1] == 0.0 ? 0.0 : __ddiv_rn(1.0, h[1]); break;
            default: v = h[2]; break;
            }
        }
        out[(long long) m * CP_MODEL_STRIDE + e] = v;
    }
}

int cpecan_hip_models_create_scaled(cpecan_ctx *c, const cpecan_sm3_model *base, const cpecan_read_scaling *scalings,
                                    int32_t n, int32_t threads, int32_t *ids) {
    if (!c || !base || !scalings || n <= 0 || !ids) return fail(CPECAN_EINVAL, "bad argument");
    if (!base->match_probs || !base->gap_x_probs || !base->gap_y_probs) return fail(CPECAN_EINVAL, "the base model has a NULL table");
    HIP_TRY(hipSetDevice(c->device));
    Lap lap("models_create_scaled");
    std::vector<double> baseRows(CP_MODEL_STRIDE);
    derive_rows(base, baseRows.data());
    PinnedBuf<double> part; /* (recycled pinned memory: no page faults, and the copy engine reads it directly) */
    HIP_TRY(part.alloc((size_t) n * CPECAN_NUM_KMERS * 3));
    int nt = threads > 0 ? threads : host_threads();
    nt = std::max(1, std::min(nt, (int) n));
    std::vector<std::thread> pool;
    for (int w = 0; w < nt; w++)
        pool.emplace_back([&, w]() {
            const double lg = -0.91893853320467267;
            for (int i = w; i < n; i += nt) {
                const cpecan_read_scaling &s = scalings[i];
                double *dst = part.p + (size_t) i * CPECAN_NUM_KMERS * 3;
                for (int k = 0; k < CPECAN_NUM_KMERS; k++) {
                    const double *a = base->match_probs + 1 + (size_t) k * CPECAN_MODEL_PARAMS;
                    const double sd = a[1] * s.var;
                    const double nmu = a[2] * s.scale_sd, lambda = a[4] * s.var_sd;
                    const double nsd = sqrt(pow(nmu, 3.0) / lambda);
                    dst[3 * k] = sd == 0.0 ? -INFINITY : lg - log(sd);
                    dst[3 * k + 1] = nsd;
                    dst[3 * k + 2] = nsd == 0.0 ? -INFINITY : lg - log(nsd);
                }
            }
        });
    for (auto &t : pool) t.join();
    lap("host libm part (threads)");
    double *fresh = nullptr;
    int rc = grow_models(c, n, &fresh);
    if (rc != CPECAN_OK) return rc;
    DevBuf<double> dBase, dScal, dPart;
    StreamFence fence{ c->prep, c->stream };
    HIP_TRY(dBase.alloc(baseRows.size()));
    HIP_TRY(dScal.alloc((size_t) n * 5));
    HIP_TRY(dPart.alloc(part.n));
    lap("device table");
    static_assert(sizeof(cpecan_read_scaling) == 5 * sizeof(double), "cpecan_read_scaling is five doubles");
    HIP_TRY(hipMemcpyAsync(dBase.p, baseRows.data(), baseRows.size() * sizeof(double), hipMemcpyHostToDevice, c->prep));
    HIP_TRY(hipMemcpyAsync(dScal.p, scalings, (size_t) n * 5 * sizeof(double), hipMemcpyHostToDevice, c->prep));
    HIP_TRY(hipMemcpyAsync(dPart.p, part.p, part.n * sizeof(double), hipMemcpyHostToDevice, c->prep));
    hipLaunchKernelGGL(cpecan_k_scale_models, dim3((unsigned) ((CP_MODEL_STRIDE + 255) / 256), (unsigned) std::min(n, 65535)),
                       dim3(256), 0, c->prep, (const double *) dBase.p, (const double *) dScal.p, (const double *) dPart.p,
                       (int) n, fresh);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->prep)); /* the staging blocks are released on return */
    lap("upload + assemble");
    for (int i = 0; i < n; i++) {
        ids[i] = c->nModels + i;
        c->switchToX.push_back(base->transitions[T_GAP_SWITCH_TO_X]);
    }
    c->nModels += n;
    return CPECAN_OK;
}

int cpecan_hip_models_download(cpecan_ctx *c, int32_t id, double *out, int64_t capacity, int64_t *nDoubles) {
    if (!c || !nDoubles) return fail(CPECAN_EINVAL, "bad argument");
    *nDoubles = CP_MODEL_STRIDE;
    if (!out) return CPECAN_OK;
    if (id < 0 || id >= c->nModels) return fail(CPECAN_EINVAL, "model id %d out of range (%d)", id, c->nModels);
    if (capacity < CP_MODEL_STRIDE) return fail(CPECAN_EINVAL, "capacity %lld < %d doubles", (long long) capacity, (int) CP_MODEL_STRIDE);
    HIP_TRY(hipSetDevice(c->device));
    if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->models.p + (size_t) id * CP_MODEL_STRIDE, CP_MODEL_STRIDE * sizeof(double), hipMemcpyDeviceToHost));
    return CPECAN_OK;
}

extern "C" __global__ void cpecan_k_set_transitions(double *models, int nModels, const double *values /* 9 + 4096 */,
                                                    int withGap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int m = blockIdx.y; m < nModels; m += gridDim.y) { /* grid.y is capped at 65535 */
        double *blk = models + (long long) m * CP_MODEL_STRIDE;
        if (i < 9) blk[i] = values[i];
        else if (withGap && i < 9 + CPECAN_NUM_KMERS)
            blk[CP_MODEL_HEADER + (long long) (i - 9) * CP_ROW + CP_GAPX] = values[i];
    }
}

int cpecan_hip_models_set_transitions(cpecan_ctx *c, const double *transitions, const double *gapX) {
    if (!c || !transitions) return fail(CPECAN_EINVAL, "bad argument");
    if (c->nModels <= 0) return fail(CPECAN_EINVAL, "the context holds no strawMan models");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<double> v(9 + CPECAN_NUM_KMERS, 0.0);
    for (int i = 0; i < 9; i++) v[(size_t) i] = transitions[i];
    if (gapX) std::copy(gapX, gapX + CPECAN_NUM_KMERS, v.begin() + 9);
    for (double &t : c->switchToX) t = transitions[T_GAP_SWITCH_TO_X];
    DevBuf<double> dv;
    StreamFence fence{ c->stream, nullptr };
    HIP_TRY(dv.alloc(v.size()));
    HIP_TRY(hipMemcpyAsync(dv.p, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(cpecan_k_set_transitions, dim3((9 + CPECAN_NUM_KMERS + 255) / 256, (unsigned) std::min(c->nModels, 65535)),
                       dim3(256), 0, c->stream, c->models.p, c->nModels, (const double *) dv.p, gapX ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream)); /* dv is released on return */
    return CPECAN_OK;
}

int cpecan_hip_selftest_division(cpecan_ctx *c, int64_t n, uint64_t seed, int64_t *mismatches) {
    if (!c || n <= 0 || !mismatches) return fail(CPECAN_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf<unsigned long long> bad;
    StreamFence fence{ c->stream, nullptr };
    HIP_TRY(bad.alloc(1));
    HIP_TRY(hipMemsetAsync(bad.p, 0, sizeof(unsigned long long), c->stream));
    if (cpecan_systolic_divtest(c->stream, n, seed, bad.p) != 0)
        return fail(CPECAN_EHIP, "division self-test launch failed");
    unsigned long long h = 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(&h, bad.p, sizeof h, hipMemcpyDeviceToHost));
    *mismatches = (int64_t) h;
    return CPECAN_OK;
}

/* Device block of one vanilla model.  Every log() the reference takes per cell
 * (stateMachine3Vanilla_cellCalculate :1391-1407, logGaussPdf :338, logInvGaussPdf :328) depends on the
 * skip bin or the k-mer only: taken here once, with the host libm the reference would call. */
static void derive_vanilla(const cpecan_vanilla_model *m, double *dst) {
    for (int i = 0; i < CP_VHDR; i++) dst[i] = 0.0;
    dst[0] = m->m_to_y_not_x;
    dst[1] = m->e_to_e;
    dst[CP_VHDR_END_M] = m->end_match_prob;
    dst[CP_VHDR_END_X] = m->end_from_x_prob;
    dst[CP_VHDR_END_Y] = m->end_from_y_prob;
    const double a_yy = m->e_to_e, a_ym = 1.0f - a_yy;
    dst[CP_VHDR_LOG_YY] = log(a_yy);
    dst[CP_VHDR_LOG_YM] = log(a_ym);
    for (int bin = 0; bin < 30; bin++) {
        const double a_mx = m->skip_probs[bin];
        const double a_my = (1 - a_mx) * m->m_to_y_not_x;
        const double a_mm = 1.0f - a_my - a_mx;
        const double a_xx = m->skip_probs[bin + 30];
        const double a_xm = 1.0f - a_xx;
        double *b = dst + CP_VHDR_BINS + bin * 5;
        b[0] = log(a_mx);
        b[1] = log(a_xx);
        b[2] = log(a_mm);
        b[3] = log(a_xm);
        b[4] = log(a_my);
    }
    const double c = -0.91893853320467267;
    double *rows = dst + CP_VHDR;
    for (int k = 0; k <= CPECAN_NUM_KMERS; k++) {
        double *r = rows + (size_t) k * CP_VROW;
        for (int t = 0; t < 2; t++) {
            double *q = r + 6 * t;
            if (k == CPECAN_NUM_KMERS) { /* not a k-mer: level -inf, the noise term kept finite */
                q[CP_V_MU] = 0.0; q[CP_V_SD] = 0.0; q[CP_V_K] = -INFINITY;
                q[CP_V_NMU] = 1.0; q[CP_V_LAMBDA] = 1.0; q[CP_V_LLAMBDA] = 0.0;
                continue;
            }
            const double *a = (t ? m->gap_y_probs : m->match_probs) + 1 + (size_t) k * CPECAN_MODEL_PARAMS;
            q[CP_V_MU] = a[0];
            q[CP_V_SD] = a[1];
            q[CP_V_K] = a[1] == 0.0 ? -INFINITY : c - log(a[1]);
            q[CP_V_NMU] = a[2];
            q[CP_V_LAMBDA] = a[4];
            q[CP_V_LLAMBDA] = log(a[4]);
        }
    }
}

int cpecan_hip_modelsv_create(cpecan_ctx *c, const cpecan_vanilla_model *models, int32_t n, int32_t threads,
                              int32_t *ids) {
    if (!c || !models || n <= 0 || !ids) return fail(CPECAN_EINVAL, "bad argument");
    for (int i = 0; i < n; i++)
        if (!models[i].match_probs || !models[i].skip_probs || !models[i].gap_y_probs)
            return fail(CPECAN_EINVAL, "model %d has a NULL table", i);
    HIP_TRY(hipSetDevice(c->device));
    const size_t old = c->hostModelsV.size();
    c->hostModelsV.resize(old + (size_t) n * CP_VMODEL_STRIDE);
    int nt = threads > 0 ? threads : host_threads();
    nt = std::max(1, std::min(nt, (int) n));
    std::vector<std::thread> pool;
    for (int w = 0; w < nt; w++)
        pool.emplace_back([&, w]() {
            for (int i = w; i < n; i += nt)
                derive_vanilla(&models[i], c->hostModelsV.data() + old + (size_t) i * CP_VMODEL_STRIDE);
        });
    for (auto &t : pool) t.join();
    for (int i = 0; i < n; i++) ids[i] = c->nModelsV + i;
    c->nModelsV += n;
    if (c->stream) (void) hipStreamSynchronize(c->stream); /* (the old table goes back to the allocator's cache) */
    hipError_t e = c->modelsV.alloc(c->hostModelsV.size());
    if (e != hipSuccess) return fail(CPECAN_EHIP, "model table allocation: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpy(c->modelsV.p, c->hostModelsV.data(), c->hostModelsV.size() * sizeof(double),
                      hipMemcpyHostToDevice));
    return CPECAN_OK;
}

int cpecan_hip_modelsh_create(cpecan_ctx *c, const cpecan_hdp_model *models, int32_t n, int32_t *ids) {
    if (!c || !models || n <= 0 || !ids) return fail(CPECAN_EINVAL, "bad argument");
    for (int i = 0; i < n; i++) {
        const cpecan_hdp_model &m = models[i];
        if (!m.alphabet || m.alphabet_size < 1 || m.alphabet_size > 16 || m.grid_length < 2 || !m.grid ||
            m.n_rows < 1 || !m.posterior_predictive || !m.spline_slopes || !m.kmer_row)
            return fail(CPECAN_EINVAL, "HDP model %d is incomplete", i);
        const std::string a(m.alphabet, (size_t) m.alphabet_size);
        if (!c->hdpAlphabet.empty() && c->hdpAlphabet != a)
            return fail(CPECAN_EINVAL, "all HDP models of a context must share one alphabet");
        c->hdpAlphabet = a;
        long long nK = 1;
        for (int q = 0; q < 6; q++) nK *= m.alphabet_size;
        for (long long k = 0; k < nK; k++)
            if (m.kmer_row[k] < 0 || m.kmer_row[k] >= m.n_rows)
                return fail(CPECAN_EINVAL, "HDP model %d: k-mer %lld points outside the tables", i, k);
    }
    HIP_TRY(hipSetDevice(c->device));
    for (int i = 0; i < n; i++) {
        const cpecan_hdp_model &m = models[i];
        long long nK = 1;
        for (int q = 0; q < 6; q++) nK *= m.alphabet_size;
        auto *t = new cpecan_ctx::HdpTables();
        c->hdpTables.push_back(t);
        const size_t cells = (size_t) m.n_rows * (size_t) m.grid_length;
        HIP_TRY(t->kmerRow.alloc((size_t) nK));
        HIP_TRY(t->grid.alloc((size_t) m.grid_length));
        HIP_TRY(t->y.alloc(cells));
        HIP_TRY(t->slope.alloc(cells));
        HIP_TRY(hipMemcpy(t->kmerRow.p, m.kmer_row, (size_t) nK * sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->grid.p, m.grid, (size_t) m.grid_length * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->y.p, m.posterior_predictive, cells * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(t->slope.p, m.spline_slopes, cells * sizeof(double), hipMemcpyHostToDevice));
        DevHdpModel d;
        for (int q = 0; q < 9; q++) d.t[q] = m.transitions[q];
        d.gridLength = m.grid_length;
        d.pad = 0;
        d.kmerRow = t->kmerRow.p;
        d.grid = t->grid.p;
        d.y = t->y.p;
        d.slope = t->slope.p;
        ids[i] = (int32_t) c->hostModelsH.size();
        c->hostModelsH.push_back(d);
    }
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    HIP_TRY(c->modelsH.alloc(c->hostModelsH.size()));
    HIP_TRY(hipMemcpy(c->modelsH.p, c->hostModelsH.data(), c->hostModelsH.size() * sizeof(DevHdpModel),
                      hipMemcpyHostToDevice));
    return CPECAN_OK;
}

int cpecan_hip_models5_create(cpecan_ctx *c, const cpecan_sm5_model *models, int32_t n, int32_t *ids) {
    if (!c || !models || n <= 0 || !ids) return fail(CPECAN_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t old = c->hostModels5.size();
    c->hostModels5.resize(old + (size_t) n * CP_MODEL5_STRIDE, 0.0);
    for (int i = 0; i < n; i++) {
        double *m = c->hostModels5.data() + old + (size_t) i * CP_MODEL5_STRIDE;
        for (int k = 0; k < 17; k++) m[k] = models[i].transitions[k];
        for (int k = 0; k < 16; k++) m[24 + k] = models[i].match_probs[k];
        for (int k = 0; k < 4; k++) m[40 + k] = models[i].gap_x_probs[k];
        for (int k = 0; k < 4; k++) m[44 + k] = models[i].gap_y_probs[k];
        ids[i] = c->nModels5 + i;
    }
    c->nModels5 += n;
    if (c->stream) (void) hipStreamSynchronize(c->stream); /* (the old table goes back to the allocator's cache) */
    hipError_t e = c->models5.alloc(c->hostModels5.size());
    if (e != hipSuccess) return fail(CPECAN_EHIP, "model table allocation: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpy(c->models5.p, c->hostModels5.data(), c->hostModels5.size() * sizeof(double),
                      hipMemcpyHostToDevice));
    return CPECAN_OK;
}

/* 4-state signal models (getStateMachine4): the strawMan rows with the machine's eleven transitions in the header.  A
 * host mirror is kept and the whole table uploaded again when models are added (a handful of models per call). */
int cpecan_hip_models4_create(cpecan_ctx *c, const cpecan_sm4_model *models, int32_t n, int32_t *ids) {
    if (!c || !models || n <= 0 || !ids) return fail(CPECAN_EINVAL, "bad argument");
    for (int i = 0; i < n; i++)
        if (!models[i].match_probs || !models[i].gap_x_probs || !models[i].gap_y_probs)
            return fail(CPECAN_EINVAL, "model %d has a NULL table", i);
    HIP_TRY(hipSetDevice(c->device));
    const size_t old = c->hostModels4.size();
    c->hostModels4.resize(old + (size_t) n * CP_MODEL_STRIDE);
    for (int i = 0; i < n; i++) {
        cpecan_sm3_model m3;
        for (int k = 0; k < 9; k++) m3.transitions[k] = models[i].transitions[k];
        m3.match_probs = models[i].match_probs;
        m3.gap_x_probs = models[i].gap_x_probs;
        m3.gap_y_probs = models[i].gap_y_probs;
        double *dst = c->hostModels4.data() + old + (size_t) i * CP_MODEL_STRIDE;
        derive_rows(&m3, dst);
        for (int k = 0; k < 11; k++) dst[k] = models[i].transitions[k];
        ids[i] = c->nModels4 + i;
    }
    if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream)); /* (the old table goes back to the allocator's cache) */
    HIP_TRY(c->models4.alloc(c->hostModels4.size()));
    HIP_TRY(hipMemcpy(c->models4.p, c->hostModels4.data(), c->hostModels4.size() * sizeof(double), hipMemcpyHostToDevice));
    c->nModels4 += n;
    return CPECAN_OK;
}

int cpecan_hip_models_clear(cpecan_ctx *c) {
    if (!c) return fail(CPECAN_EINVAL, "ctx is NULL");
    (void) hipSetDevice(c->device);
    if (c->stream) (void) hipStreamSynchronize(c->stream); /* the tables go back to the allocator's cache: no reader may be left */
    c->modelEpoch++; /* batches created before this call hold ids into tables that are gone: batch_run refuses them */
    c->models.release();
    c->switchToX.clear();
    c->nModels = 0;
    c->models5.release();
    c->hostModels5.clear();
    c->nModels5 = 0;
    c->modelsV.release();
    c->hostModelsV.clear();
    c->nModelsV = 0;
    c->models4.release();
    c->hostModels4.clear();
    c->nModels4 = 0;
    for (auto *t : c->hdpTables) delete t;
    c->hdpTables.clear();
    c->hostModelsH.clear();
    c->modelsH.release();
    c->hdpAlphabet.clear();
    return CPECAN_OK;
}

int cpecan_hip_batch_destroy(cpecan_batch *b) {
    if (!b) return CPECAN_OK;
    /* the batch keeps its own device id: a caller (a garbage collector, say) may destroy the context first, and
     * nothing here may depend on it then */
    (void) hipSetDevice(b->device);
    Lap lap("batch_destroy");
    /* the batch's device memory goes back to the allocator's cache, not to the driver (which would wait for the
     * device): nothing of this batch may still be running when another batch is handed the blocks */
    if (b->ev2 && b->ran) (void) hipEventSynchronize(b->ev2);
    if (b->ev0) (void) hipEventDestroy(b->ev0);
    if (b->ev1) (void) hipEventDestroy(b->ev1);
    if (b->ev2) (void) hipEventDestroy(b->ev2);
    for (hipEvent_t e : b->evStage) (void) hipEventDestroy(e);
    for (hipEvent_t e : b->evJoin) (void) hipEventDestroy(e);
    if (b->evFork) (void) hipEventDestroy(b->evFork);
    if (b->gStreamOwned)
        for (hipStream_t st : b->gStream) (void) hipStreamDestroy(st);
    for (hipStream_t st : b->gStreamB) (void) hipStreamDestroy(st);
    if (b->asmPost) (void) hipStreamDestroy(b->asmPost);
    for (hipEvent_t e : b->evPost) (void) hipEventDestroy(e);
    if (b->hPacked) pinned_cache().put(b->hPacked, b->hPackedBlock);
    if (b->hPost) pinned_cache().put(b->hPost, b->hPostBlock);
    if (b->hUndecided) pinned_cache().put(b->hUndecided, b->hUndecidedBlock);
    delete b;
    (void) hipGetLastError(); /* a failed clean-up call must not surface as the "last error" of a later launch */
    return CPECAN_OK;
}

#define B_TRY(expr)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            int rc_ = fail(CPECAN_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                              \
            cpecan_hip_batch_destroy(b);                                                     \
            return rc_;                                                                      \
        }                                                                                    \
    } while (0)

/* events != NULL: k-mers against events with a 3-state signal model; yChars != NULL: DNA against DNA
 * with a 5-state symbol model (nEvents then counts the bases of yChars) */
static int batch_create_impl(cpecan_ctx *c, const cpecan_item *items, int64_t nItems,
                             const char *xChars, int64_t nX, const double *events, const char *yChars,
                             int64_t nEvents, const int64_t *anchors, int64_t nAnchorPairs,
                             const cpecan_band_params *params, int32_t mode, int32_t kernel,
                             int32_t flags, cpecan_batch **out, bool vanilla = false, bool hdp = false, bool sm4 = false) {
    if (sm4 && (mode != CPECAN_MODE_POSTERIOR || (flags & CPECAN_FLAG_DEBUG_DUMP)))
        return fail(CPECAN_EINVAL, "4-state batches: posterior decode only, no cell dumps");
    const bool dna = yChars != nullptr;
    if (hdp && (flags & CPECAN_FLAG_DEBUG_DUMP)) return fail(CPECAN_EINVAL, "HDP batches: no cell dumps");
    if (hdp && mode == CPECAN_MODE_EXPECTATIONS && (flags & CPECAN_FLAG_UNBANDED))
        return fail(CPECAN_EINVAL, "expectations run over the banded matrix only");
    if (vanilla && (flags & CPECAN_FLAG_DEBUG_DUMP)) return fail(CPECAN_EINVAL, "vanilla batches: no cell dumps");
    if (vanilla && mode == CPECAN_MODE_EXPECTATIONS && (flags & CPECAN_FLAG_UNBANDED))
        return fail(CPECAN_EINVAL, "expectations run over the banded matrix only");
    const int S = dna ? 5 : sm4 ? 4 : 3; /* states per cell */
    if (!c || !items || nItems <= 0 || !xChars || (!events && !yChars) || !params || !out)
        return fail(CPECAN_EINVAL, "bad argument");
    if (dna && (flags & CPECAN_FLAG_DEBUG_DUMP)) return fail(CPECAN_EINVAL, "DNA batches: no cell dumps");
    if (dna && mode == CPECAN_MODE_EXPECTATIONS && (flags & CPECAN_FLAG_UNBANDED))
        return fail(CPECAN_EINVAL, "expectations run over the banded matrix only");
    if (nAnchorPairs > 0 && !anchors) return fail(CPECAN_EINVAL, "anchors is NULL");
    if (mode != CPECAN_MODE_POSTERIOR && mode != CPECAN_MODE_EXPECTATIONS)
        return fail(CPECAN_EINVAL, "unknown mode %d", mode);
    if (params->diagonalExpansion < 0 || (params->diagonalExpansion & 1) ||
        params->traceBackDiagonals < 1 || params->minDiagsBetweenTraceBack < 2 ||
        params->traceBackDiagonals + 1 >= params->minDiagsBetweenTraceBack)
        return fail(CPECAN_EINVAL, "banding parameters violate the prerequisites of "
                                   "getPosteriorProbsWithBanding (pairwiseAligner.c:880-884)");
    *out = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    const bool unbanded = (flags & CPECAN_FLAG_UNBANDED) != 0;
    if (unbanded && (mode != CPECAN_MODE_POSTERIOR || kernel == CPECAN_KERNEL_SYSTOLIC))
        return fail(CPECAN_EINVAL, "un-banded alignment: posterior mode on the general kernel only");

    /* per-item validation + band tables (host integer work): offsets in one serial pass, then the items are dealt
     * to the host threads (band, cell prefix, traceback schedule: ~15 000 diagonals per C3 read) */
    std::vector<DevItem> hItems((size_t) nItems);
    std::vector<int, NoInit<int>> hL, hR;
    std::vector<long long, NoInit<long long>> hPre;
    long long cellTotal = 0, pairTotal = 0, totTotal = 0, bwsTotal = 0, trackTotal = 0;
    int globalMaxWidth = 0, maxSpan = 1, maxLX = 0, maxWindows = 0;
    long long maxLXY = 0; /* the longest sequence of the batch, either side */
    bool systolicOk = true; /* band edges move by at most one k-mer per diagonal */
    std::vector<long long> hTrackBase((size_t) nItems);
    Lap lap("batch_create");
    long long diagTotal = 0;
    for (int64_t i = 0; i < nItems; i++) {
        const cpecan_item &s = items[i];
        if (s.lX < 0 || s.lY < 0 || s.x_offset < 0 || s.y_offset < 0 || s.n_anchors < 0 ||
            s.anchor_offset < 0 || s.x_offset + s.lX + (!dna && s.lX > 0 ? 5 : 0) > nX ||
            s.y_offset + s.lY > nEvents || s.anchor_offset + s.n_anchors > nAnchorPairs)
            return fail(CPECAN_EINVAL, "item %lld points outside the supplied buffers", (long long) i);
        if (s.model_id < 0 || s.model_id >= (dna ? c->nModels5 : vanilla ? c->nModelsV : sm4 ? c->nModels4
                                                      : hdp ? (int) c->hostModelsH.size() : c->nModels))
            return fail(CPECAN_EINVAL, "item %lld: unknown model id %d", (long long) i, s.model_id);
        if (s.lX + s.lY >= (1ll << 30)) return fail(CPECAN_EINVAL, "item %lld too long", (long long) i);
        DevItem &d = hItems[(size_t) i];
        d.lX = s.lX; d.lY = s.lY; d.xOff = s.x_offset; d.yOff = s.y_offset;
        d.anchorOff = s.anchor_offset; d.nAnchors = s.n_anchors;
        d.model = s.model_id; d.raggedL = s.ragged_left ? 1 : 0; d.raggedR = s.ragged_right ? 1 : 0;
        d.diagBase = diagTotal;
        diagTotal += s.lX + s.lY + 1;
    }
    struct ItemStats {
        int maxSpan = 1, windows = 0, badItem = -1, badRc = 0;
        bool systolicOk = true;
    };
    /* The band of every item: as matrix columns (first, last) per diagonal -- what the register-resident kernels
     * read -- written straight into a pinned block; the x-y intervals and the cell prefix sums of the general kernel
     * only when the batch will (or, on the second call, turns out to) run on it.  A thread's working copy of one
     * item's intervals stays in its cache. */
    PinnedBuf<int> hTab;
    /* the assembly sweeps' plan (build_asm_plan), for the batches that can run on them */
    const bool wantPlan = !dna && !vanilla && !hdp && !sm4 && !unbanded && mode == CPECAN_MODE_POSTERIOR &&
                          kernel != CPECAN_KERNEL_GENERAL && use_wave_kernels() && !(flags & CPECAN_FLAG_WORKGROUP_KERNELS) &&
                          !(flags & CPECAN_FLAG_DEBUG_DUMP);
    std::vector<std::vector<AsmPlanWin>> planWins(wantPlan ? (size_t) nItems : 0);
    std::vector<long long> hPlanOff(wantPlan ? (size_t) nItems : 0);
    PinnedBuf<AsmPlanCtl> hCtl;
    StreamFence prepFence{ c->prep, nullptr }; /* (hTab, hCtl and the plan records below are uploaded through it) */
    long long ctlTotal = 0;
    if (wantPlan) {
        for (int64_t i = 0; i < nItems; i++) {
            hPlanOff[(size_t) i] = ctlTotal;
            ctlTotal += (items[i].lX + items[i].lY) / ASM_BLOCK + 2;
        }
        HIP_TRY(hCtl.alloc((size_t) ctlTotal));
    }
    bool keptGeneral = false;
    long long maxDiags = 0;
    for (int64_t i = 0; i < nItems; i++) maxDiags = std::max<long long>(maxDiags, items[i].lX + items[i].lY + 1);
    auto build_bands = [&](bool general) -> int {
        if (general) {
            hL.resize((size_t) diagTotal);
            hR.resize((size_t) diagTotal);
            hPre.resize((size_t) diagTotal);
            keptGeneral = true;
        }
        maxSpan = 1;
        maxWindows = 0;
        systolicOk = true;
        const int nt = (int) std::min<int64_t>(diagTotal > 2000000 ? host_threads() : 1, nItems);
        std::vector<ItemStats> stats((size_t) nt);
        auto work = [&](int w) {
            ItemStats &st = stats[(size_t) w];
            std::vector<int, NoInit<int>> own(general ? 0 : 2 * (size_t) maxDiags);
            for (int64_t i = w; i < nItems; i += nt) {
                const cpecan_item &s = items[i];
                DevItem &d = hItems[(size_t) i];
                const long long nDiag = s.lX + s.lY + 1;
                int *Lp = general ? hL.data() + d.diagBase : own.data();
                int *Rp = general ? hR.data() + d.diagBase : own.data() + maxDiags;
                /* getAlignedPairsWithoutBanding builds its band from no anchors, expansion 2 (:1532) */
                int rc = cpecan_band_construct(unbanded || !anchors ? nullptr : anchors + 2 * s.anchor_offset,
                                               unbanded ? 0 : s.n_anchors, s.lX, s.lY,
                                               unbanded ? 2 : params->diagonalExpansion, Lp, Rp);
                if (rc != CPECAN_OK) {
                    if (st.badItem < 0) { st.badItem = (int) i; st.badRc = rc; }
                    continue;
                }
                long long cells = 0;
                int maxW = 0;
                long long *pre = general ? hPre.data() + d.diagBase : nullptr;
                int *tab = hTab.p + d.diagBase * 2;
                /* traceback schedule of getPosteriorProbsWithBanding (:917-918): longest span of forward
                 * diagonals that must be resident at once, and the edge-step property the register-resident
                 * kernels rely on */
                long long tracedBackTo = 0;
                int windows = 0, pmn = 0, pmx = 0;
                for (long long k = 0; k < nDiag; k++) {
                    if (pre) pre[k] = cells;
                    const int wd = ((Rp[k] - Lp[k]) >> 1) + 1;
                    cells += wd;
                    maxW = std::max(maxW, wd);
                    const int xmn = (int) ((k + Lp[k]) / 2), xmx = (int) ((k + Rp[k]) / 2);
                    tab[k * 2] = xmn;
                    tab[k * 2 + 1] = xmx;
                    if (k >= 1) {
                        if (xmn < pmn || xmn > pmn + 1 || xmx < pmx || xmx > pmx + 1) st.systolicOk = false;
                        const bool atEnd = k == nDiag - 1;
                        const bool tb = k >= tracedBackTo + params->minDiagsBetweenTraceBack &&
                                        wd <= params->diagonalExpansion * 2 + 1;
                        if (atEnd || tb) {
                            windows++;
                            st.maxSpan = (int) std::max<long long>(st.maxSpan, k - tracedBackTo + 1);
                            tracedBackTo = k - (params->traceBackDiagonals + 1);
                        }
                    }
                    pmn = xmn;
                    pmx = xmx;
                }
                d.nCells = cells;
                d.maxWidth = maxW;
                st.windows = std::max(st.windows, windows);
                if (wantPlan) build_asm_plan(tab, nDiag, *params, planWins[(size_t) i], hCtl.p + hPlanOff[(size_t) i]);
            }
        };
        if (nt <= 1) work(0);
        else {
            std::vector<std::thread> pool;
            for (int w = 0; w < nt; w++) pool.emplace_back(work, w);
            for (auto &t : pool) t.join();
        }
        int bad = -1, badRc = 0;
        for (const ItemStats &st : stats) {
            maxSpan = std::max(maxSpan, st.maxSpan);
            maxWindows = std::max(maxWindows, st.windows);
            systolicOk = systolicOk && st.systolicOk;
            if (st.badItem >= 0 && (bad < 0 || st.badItem < bad)) { bad = st.badItem; badRc = st.badRc; }
        }
        if (bad >= 0) return fail(badRc, "item %lld: anchors do not describe a valid band", (long long) bad);
        return CPECAN_OK;
    };
    /* (what the kernel choice below will come to, as far as it is known before the bands are) */
    const bool surelyGeneral = dna || sm4 || kernel == CPECAN_KERNEL_GENERAL || unbanded || (flags & CPECAN_FLAG_DEBUG_DUMP) ||
                               ((hdp || vanilla) && (flags & CPECAN_FLAG_GENERAL_KERNEL));
    HIP_TRY(hTab.alloc((size_t) diagTotal * 2 + 2));
    {
        int rc = build_bands(surelyGeneral);
        if (rc != CPECAN_OK) return rc;
    }
    for (int64_t i = 0; i < nItems; i++) {
        const cpecan_item &s = items[i];
        DevItem &d = hItems[(size_t) i];
        const long long nDiag = s.lX + s.lY + 1;
        hTrackBase[(size_t) i] = trackTotal;
        trackTotal += s.lX + 1;
        maxLX = std::max<int>(maxLX, (int) s.lX);
        maxLXY = std::max<long long>(maxLXY, std::max<long long>(s.lX, s.lY));
        globalMaxWidth = std::max(globalMaxWidth, d.maxWidth);
        d.cellBase = cellTotal;
        cellTotal += d.nCells;
        d.pairBase = pairTotal;
        /* the HDP machine scores with linear densities (quirk Q6): its posteriors are flat and far more
         * cells pass the threshold (2887 pairs for a ~800-event read in the reference's own test); a batch whose
         * counts outgrow this first guess is re-run with the counted sizes (ensure_counts) */
        d.pairCap = (hdp ? 16 : 4) * (s.lX + s.lY) + 64;
        pairTotal += d.pairCap;
        d.totBase = totTotal;
        d.totCap = (nDiag + 9) / 10 + nDiag / std::max<long long>(1, params->minDiagsBetweenTraceBack -
                                                                     params->traceBackDiagonals - 1) + 4;
        totTotal += d.totCap;
        d.bwsBase = bwsTotal;
        bwsTotal += 3ll * d.maxWidth * S;
    }

    lap("band construction and window schedule (host)");
    cpecan_batch *b = new (std::nothrow) cpecan_batch();
    if (!b) return fail(CPECAN_EINVAL, "out of host memory");
    b->ctx = c;
    b->device = c->device;
    b->modelEpoch = c->modelEpoch;
    b->nItems = nItems;
    b->mode = mode;
    b->flags = flags;
    b->compactPairs = maxLXY < 65536;
    b->nModels = dna ? c->nModels5 : vanilla ? c->nModelsV : sm4 ? c->nModels4 : hdp ? (int) c->hostModelsH.size() : c->nModels;
    b->expectLen = dna ? CPECAN_EXPECTATION5_LEN : vanilla ? CPECAN_EXPECTATIONV_LEN
                   : hdp ? CPECAN_EXPECTATIONH_LEN : CPECAN_EXPECTATION_LEN;
    b->P.threshold = params->threshold;
    b->P.minDiags = params->minDiagsBetweenTraceBack;
    b->P.tbDiags = params->traceBackDiagonals;
    b->P.expansion = params->diagonalExpansion;
    b->P.mode = mode;
    b->P.debug = (flags & CPECAN_FLAG_DEBUG_DUMP) ? 1 : 0;
    b->P.unbanded = unbanded ? 1 : 0;
    b->P.scanDecode = (flags & CPECAN_FLAG_SCAN_DECODE) ? 1 : 0;
    b->P.logThrSlack = params->threshold > 0.0 ? log(params->threshold) - 1e-3 : -INFINITY;
    b->P.ldsWidth = 0; /* (set per launch by the kernels that use it) */

    /* the HDP and vanilla machines have wave-per-alignment kernels of their own, for the posterior decode and for
     * the E-step (the 5-state machine runs on the general kernel); CPECAN_FLAG_GENERAL_KERNEL keeps such a batch on
     * the general kernel */
    const bool machineWave = (hdp || vanilla) && !(flags & CPECAN_FLAG_GENERAL_KERNEL);
    int useKernel = dna || sm4 || ((hdp || vanilla) && !machineWave) ? CPECAN_KERNEL_GENERAL
                    : hdp || vanilla ? CPECAN_KERNEL_AUTO : kernel;
    /* the builds of the register-resident kernels this batch would run on, and the widest band they take */
    const SyBuild *fam = hdp ? HV_BUILDS : vanilla ? VV_BUILDS
                         : (use_wave_kernels() && !(flags & CPECAN_FLAG_WORKGROUP_KERNELS)) ? WV_BUILDS : SY_BUILDS;
    const int famMaxWidth = fam[3].max_width();
    b->dna = dna;
    b->vanilla = vanilla;
    b->hdp = hdp;
    b->sm4 = sm4;
    if (useKernel == CPECAN_KERNEL_AUTO)
        useKernel = (globalMaxWidth <= famMaxWidth && systolicOk && !b->P.debug && !unbanded)
                        ? CPECAN_KERNEL_SYSTOLIC : CPECAN_KERNEL_GENERAL;
    if (useKernel == CPECAN_KERNEL_SYSTOLIC && (globalMaxWidth > famMaxWidth || !systolicOk)) {
        delete b;
        return fail(CPECAN_EINVAL, "band is %d cells wide (systolic kernel: at most %d, edges moving "
                    "one k-mer per diagonal)", globalMaxWidth, famMaxWidth);
    }
    if (useKernel == CPECAN_KERNEL_SYSTOLIC && b->P.debug) {
        delete b;
        return fail(CPECAN_EINVAL, "cell dumps are only available from the general kernel");
    }
    b->kernel = useKernel;
    b->maxWidth = globalMaxWidth;
    /* the build with the fewest waves per workgroup whose slots hold the widest band: the fewer waves an alignment
     * takes, the more alignments a CU holds (CPECAN_SYSTOLIC_ROWS=N asks for at least N waves: tests, timing) */
    {
        const char *rows = getenv("CPECAN_SYSTOLIC_ROWS");
        int r = rows ? atoi(rows) : 1;
        r = r < 1 ? 1 : r > 4 ? 4 : r;
        while (r < 4 && globalMaxWidth > fam[r - 1].max_width()) r++;
        b->sy = &fam[r - 1];
        b->trackRow = vanilla ? cpecan_wave_track_row_doubles_vanilla()
                      : b->sy->wave ? cpecan_wave_track_row_doubles() : CP_ROW;
    }
    b->hItems = hItems;

    B_TRY(b->items.alloc((size_t) nItems));
    B_TRY(hipMemcpyAsync(b->items.p, hItems.data(), (size_t) nItems * sizeof(DevItem), hipMemcpyHostToDevice, c->prep));
    B_TRY(b->chars.alloc((size_t) nX + 8));
    B_TRY(hipMemsetAsync(b->chars.p, 0, (size_t) nX + 8, c->prep));
    B_TRY(hipMemcpyAsync(b->chars.p, xChars, (size_t) nX, hipMemcpyHostToDevice, c->prep));
    B_TRY(b->kidx.alloc((size_t) nX + 8));
    if (dna) {
        B_TRY(b->charsY.alloc((size_t) nEvents + 8));
        B_TRY(hipMemsetAsync(b->charsY.p, 0, (size_t) nEvents + 8, c->prep));
        B_TRY(hipMemcpyAsync(b->charsY.p, yChars, (size_t) nEvents, hipMemcpyHostToDevice, c->prep));
    } else {
        B_TRY(b->events.alloc((size_t) 3 * nEvents + 8));
        if (vanilla) {
            /* the batch's own copy of the events carries log(noise) (host libm, :325) in place of the duration, which
             * nothing on the device reads: the wave kernels stage events from this one array */
            std::vector<double, NoInit<double>> ev3((size_t) 3 * nEvents);
            for (int64_t i = 0; i < nEvents; i++) {
                ev3[(size_t) 3 * i] = events[3 * i];
                ev3[(size_t) 3 * i + 1] = events[3 * i + 1];
                ev3[(size_t) 3 * i + 2] = log(events[3 * i + 1]);
            }
            B_TRY(hipMemcpyAsync(b->events.p, ev3.data(), (size_t) 3 * nEvents * sizeof(double), hipMemcpyHostToDevice, c->prep));
            B_TRY(hipStreamSynchronize(c->prep)); /* ev3 ends here */
        } else
            B_TRY(hipMemcpyAsync(b->events.p, events, (size_t) 3 * nEvents * sizeof(double), hipMemcpyHostToDevice, c->prep));
        if (vanilla) { /* emissions_signal_logInvGaussPdf takes log(eventNoise) per cell (:325) */
            std::vector<double> ln((size_t) nEvents + 1);
            for (int64_t i = 0; i < nEvents; i++) ln[(size_t) i] = log(events[3 * i + 1]);
            B_TRY(b->logNoise.alloc((size_t) nEvents + 8));
            B_TRY(hipMemcpyAsync(b->logNoise.p, ln.data(), (size_t) nEvents * sizeof(double), hipMemcpyHostToDevice, c->prep));
            B_TRY(hipStreamSynchronize(c->prep)); /* ln ends here */
        }
    }
    B_TRY(hipStreamSynchronize(c->prep));
    lap("upload sequences and events");
    /* (the anchors stay on the host: the bands they describe were built there, above) */
    B_TRY(b->pairs.alloc((size_t) pairTotal * 3));
    B_TRY(b->pairLogp.alloc((size_t) pairTotal));
    B_TRY(b->nPairs.alloc((size_t) nItems));
    B_TRY(b->nTot.alloc((size_t) nItems));
    B_TRY(b->nCells.alloc((size_t) nItems));
    B_TRY(b->totXay.alloc((size_t) totTotal));
    B_TRY(b->totVal.alloc((size_t) totTotal));
    B_TRY(b->expect.alloc((size_t) std::max(b->nModels, 1) * b->expectLen));
    B_TRY(hipMemsetAsync(b->expect.p, 0, b->expect.n * sizeof(double), c->prep));
    b->hNCells.resize((size_t) nItems);
    for (int64_t i = 0; i < nItems; i++) b->hNCells[(size_t) i] = hItems[(size_t) i].nCells;
    B_TRY(hipStreamSynchronize(c->prep));
    lap("output buffers");

    if (useKernel == CPECAN_KERNEL_GENERAL) {
        if (!keptGeneral) { /* the band turned out too wide (or too ragged) for the register-resident kernels */
            int rc = build_bands(true);
            if (rc != CPECAN_OK) {
                cpecan_hip_batch_destroy(b);
                return rc;
            }
        }
        B_TRY(b->bandL.alloc(hL.size()));
        B_TRY(b->bandR.alloc(hR.size()));
        B_TRY(b->cellPrefix.alloc(hPre.size()));
        B_TRY(hipMemcpyAsync(b->bandL.p, hL.data(), hL.size() * sizeof(int), hipMemcpyHostToDevice, c->prep));
        B_TRY(hipMemcpyAsync(b->bandR.p, hR.data(), hR.size() * sizeof(int), hipMemcpyHostToDevice, c->prep));
        B_TRY(hipMemcpyAsync(b->cellPrefix.p, hPre.data(), hPre.size() * sizeof(long long), hipMemcpyHostToDevice, c->prep));
        B_TRY(b->Fstore.alloc((size_t) cellTotal * S));
        B_TRY(b->Bstore.alloc((size_t) bwsTotal));
        if (b->P.debug) {
            B_TRY(b->dbgB.alloc((size_t) cellTotal * 3));
            B_TRY(hipMemsetAsync(b->dbgB.p, 0xff, (size_t) cellTotal * 3 * sizeof(double), c->prep));
        }
    } else {
        /* one workgroup per alignment and launch; the ring of forward diagonals lives per alignment
         * because the forward and backward kernels of a window are separate launches */
        b->nWorkers = (int) nItems;
        b->nWindows = maxWindows;
        b->ringD = 64;
        /* the kernels mask with ringD-1.  The wave kernels sweep window w back while the forward sweep of window w+1
         * is writing: the ring holds two windows -- three where the assembly sweeps may run (decided below; the same
         * conditions but for what is not known yet): there the forward sweep of window w+2 does not wait for the totals
         * and the decode of window w, whose re-sweep kernel may still read that window's rows */
        const bool asmOffEarly = getenv("CPECAN_ASM") != nullptr && atoi(getenv("CPECAN_ASM")) == 0;
        const bool asmLikely = wantPlan && !asmOffEarly && b->sy->wave && b->sy->rows == ASM_L && globalMaxWidth <= ASM_MAX_WIDTH;
        const int ringWindows = !b->sy->wave || maxWindows <= 1 ? 1
                                : asmLikely && maxWindows > 2 && !(flags & CPECAN_FLAG_SMALL_FOOTPRINT) ? 3 : 2;
        while (b->ringD < (ringWindows > 1 ? ringWindows * maxSpan + 8 : maxSpan + 4)) b->ringD *= 2;
        /* the wave kernels keep one more row behind the ring: the -inf row lanes without a cell read */
        b->ringDoubles = (long long) (b->ringD + (b->sy->wave ? 1 : 0)) * b->sy->ring_row_doubles();
        if (getenv("CPECAN_RING_PAD")) b->ringDoubles += atoll(getenv("CPECAN_RING_PAD"));
        b->maxLX = maxLX;
        B_TRY(b->Fstore.alloc((size_t) nItems * (size_t) b->ringDoubles));
        lap("ring allocation");
        /* the band as matrix columns per diagonal */
        B_TRY(b->bandTab.alloc((size_t) diagTotal * 2 + 2));
        B_TRY(hipMemcpyAsync(b->bandTab.p, hTab.p, (size_t) diagTotal * 2 * sizeof(int), hipMemcpyHostToDevice, c->prep));
        {
            const char *g = getenv("CPECAN_SYSTOLIC_GROUPS");
            int G = g ? atoi(g) : b->sy->wave ? 1 : 2;
            if (G < 1) G = 1;
            if (G > 8) G = 8;
            if ((int64_t) G > nItems) G = (int) nItems;
            b->nGroups = G;
            b->gStream.assign((size_t) G, nullptr);
            b->evJoin.assign((size_t) G, nullptr);
            /* a wave batch of one group sweeps forward on the context's own stream: two streams per batch, so that
             * two batches in flight stay within the four hardware queues a process gets */
            b->gStreamOwned = !(b->sy->wave && G == 1);
            if (b->gStreamOwned)
                for (auto &st : b->gStream) B_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            else
                b->gStream[0] = c->stream;
            if (b->sy->wave) {
                b->gStreamB.assign((size_t) G, nullptr);
                for (auto &st : b->gStreamB) B_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            }
            for (auto &e : b->evJoin) B_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            B_TRY(hipEventCreateWithFlags(&b->evFork, hipEventDisableTiming));
        }
        B_TRY(hipStreamSynchronize(c->prep));
        lap("band table upload, streams");
        if (mode == CPECAN_MODE_EXPECTATIONS)
            B_TRY(b->Bring.alloc((size_t) nItems * (size_t) b->ringD * (size_t) b->sy->bring_row_doubles()));
        b->stateBytes = b->sy->wave ? cpecan_wave_state_bytes() : cpecan_systolic_state_bytes();
        B_TRY(b->syStates.alloc((size_t) nItems * (size_t) b->stateBytes));
        b->scratchBytes = (b->sy->scratch_bytes(b->ringD) + 63) / 64 * 64;
        B_TRY(b->syScratch.alloc((size_t) nItems * (size_t) b->scratchBytes));
        B_TRY(b->track.alloc((size_t) trackTotal * (size_t) b->trackRow));
        B_TRY(b->trackBase.alloc((size_t) nItems));
        B_TRY(hipMemcpyAsync(b->trackBase.p, hTrackBase.data(), (size_t) nItems * sizeof(long long),
                             hipMemcpyHostToDevice, c->prep));
        B_TRY(hipStreamSynchronize(c->prep));
        lap("state, scratch, track allocation");
        /* the hand-scheduled assembly sweeps take the strawMan machine's posterior batches whose bands need three cells
         * per lane and fit their staging scheme (CPECAN_ASM=0: the compiled kernels, for tests and timing) */
        const bool asmOff = getenv("CPECAN_ASM") != nullptr && atoi(getenv("CPECAN_ASM")) == 0; /* (read per batch) */
        if (wantPlan && !asmOff && b->sy->wave && b->sy->rows == ASM_L && globalMaxWidth <= ASM_MAX_WIDTH && b->nGroups == 1 &&
            b->stateBytes == (int) sizeof(WvState) && b->sy->ring_row_doubles() * 8 == ASM_ROW_BYTES /* (one ring format) */ &&
            cpecan_asm_load(c->device) == 0) {
            b->asmMaxWindows = std::max(maxWindows, 1);
            PinnedBuf<AsmPlanWin> hWin;
            StreamFence winFence{ c->prep, nullptr };
            B_TRY(hWin.alloc((size_t) nItems * (size_t) b->asmMaxWindows));
            memset(hWin.p, 0, (size_t) nItems * (size_t) b->asmMaxWindows * sizeof(AsmPlanWin));
            for (int64_t i = 0; i < nItems; i++)
                std::copy(planWins[(size_t) i].begin(), planWins[(size_t) i].end(), hWin.p + (size_t) i * (size_t) b->asmMaxWindows);
            B_TRY(b->planWin.alloc(hWin.n));
            B_TRY(b->planCtl.alloc((size_t) ctlTotal));
            B_TRY(b->planOff.alloc((size_t) nItems));
            B_TRY(b->asmCtx.alloc((size_t) nItems * 3 * ASM_CTX_BYTES));
            B_TRY(b->asmMasks.alloc((size_t) (diagTotal + 2) * (ASM_MASK_BYTES / 4)));
            B_TRY(hipMemsetAsync(b->asmMasks.p + (size_t) diagTotal * (ASM_MASK_BYTES / 4), 0, 2 * ASM_MASK_BYTES, c->prep));
            B_TRY(hipMemcpyAsync(b->planWin.p, hWin.p, hWin.n * sizeof(AsmPlanWin), hipMemcpyHostToDevice, c->prep));
            B_TRY(hipMemcpyAsync(b->planCtl.p, hCtl.p, (size_t) ctlTotal * sizeof(AsmPlanCtl), hipMemcpyHostToDevice, c->prep));
            B_TRY(hipMemcpyAsync(b->planOff.p, hPlanOff.data(), (size_t) nItems * sizeof(long long), hipMemcpyHostToDevice, c->prep));
            if (cpecan_asm_launch_masks(c->prep, b->items.p, nItems, maxDiags, b->bandTab.p, b->asmMasks.p) != 0 ||
                cpecan_asm_launch_ctx_init(c->prep, b->items.p, nItems, b->asmCtx.p, ASM_CTX_BYTES, b->Fstore.p, b->ringDoubles,
                                           b->ringD) != 0) {
                cpecan_hip_batch_destroy(b);
                return fail(CPECAN_EHIP, "context kernel launch failed: %s", hipGetErrorString(hipGetLastError()));
            }
            B_TRY(hipStreamSynchronize(c->prep)); /* hWin ends here */
            if (b->ringD >= 3 * maxSpan + 8 || maxWindows <= 2) {
                /* the post kernel of a window runs beside the next window's sweeps (batch_run): the sweep back of window
                 * w+1 fills one half of the scratch while the post kernel of window w reads the other */
                B_TRY(b->syScratch.alloc(2 * (size_t) nItems * (size_t) b->scratchBytes));
                B_TRY(hipStreamCreateWithFlags(&b->asmPost, hipStreamNonBlocking));
                b->evPost.assign((size_t) b->asmMaxWindows, nullptr);
                for (auto &e : b->evPost) B_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
            b->useAsm = true;
            /* CPECAN_ASM=1: the forward sweep only (the compiled sweep back reads what it writes: tests, timing) */
            b->asmBackward = !(getenv("CPECAN_ASM") != nullptr && atoi(getenv("CPECAN_ASM")) == 1);
            if (getenv("CPECAN_ASM_TRACE"))
                fprintf(stderr, "[cpecan asm] batch of %lld alignments, widest band %d, %d windows: assembly sweeps\n",
                        (long long) nItems, globalMaxWidth, b->asmMaxWindows);
            lap("assembly sweeps: plan upload, contexts");
        }
    }
    B_TRY(hipEventCreate(&b->ev0));
    B_TRY(hipEventCreate(&b->ev1));
    B_TRY(hipEventCreate(&b->ev2));

    if (hdp) { /* k-mer ids over the HDP's alphabet, once per batch like the k-mer indices below */
        unsigned long long lo = 0, hi = 0;
        for (size_t q = 0; q < c->hdpAlphabet.size(); q++) {
            const unsigned long long ch = (unsigned char) c->hdpAlphabet[q];
            if (q < 8) lo |= ch << (8 * q);
            else hi |= ch << (8 * (q - 8));
        }
        B_TRY(b->kid.alloc((size_t) nX + 8));
        const int blocks = (int) ((nX + 255) / 256);
        if (blocks > 0)
            hipLaunchKernelGGL(cpecan_k_hdp_kmer_id, dim3(blocks), dim3(256), 0, c->prep,
                               (const char *) b->chars.p, (long long) nX, lo, hi, (int) c->hdpAlphabet.size(),
                               b->kid.p);
        B_TRY(hipGetLastError());
        B_TRY(hipStreamSynchronize(c->prep));
    }
    /* k-mer indices are part of input preparation (done once, like H2D) */
    if (!dna && !hdp) {
        long long n = (long long) nX;
        int threads = 256;
        int blocks = (int) ((n + threads - 1) / threads);
        if (blocks > 0)
            hipLaunchKernelGGL(cpecan_k_kmer_index, dim3(blocks), dim3(threads), 0, c->prep,
                               (const char *) b->chars.p, n, b->kidx.p);
        B_TRY(hipGetLastError());
        B_TRY(hipStreamSynchronize(c->prep));
    }
    B_TRY(hipStreamSynchronize(c->prep)); /* every upload above has landed */
    lap("k-mer index kernel");
    *out = b;
    return CPECAN_OK;
}

int cpecan_hip_batch_create(cpecan_ctx *c, const cpecan_item *items, int64_t nItems,
                            const char *xChars, int64_t nX, const double *events, int64_t nEvents,
                            const int64_t *anchors, int64_t nAnchorPairs,
                            const cpecan_band_params *params, int32_t mode, int32_t kernel,
                            int32_t flags, cpecan_batch **out) {
    if (!events) return fail(CPECAN_EINVAL, "bad argument");
    return batch_create_impl(c, items, nItems, xChars, nX, events, nullptr, nEvents, anchors, nAnchorPairs,
                             params, mode, kernel, flags, out);
}

int cpecan_hip_batch_create_vanilla(cpecan_ctx *c, const cpecan_item *items, int64_t nItems,
                                    const char *xChars, int64_t nX, const double *events, int64_t nEvents,
                                    const int64_t *anchors, int64_t nAnchorPairs,
                                    const cpecan_band_params *params, int32_t flags, cpecan_batch **out) {
    if (!events) return fail(CPECAN_EINVAL, "bad argument");
    return batch_create_impl(c, items, nItems, xChars, nX, events, nullptr, nEvents, anchors, nAnchorPairs, params,
                             (flags & CPECAN_FLAG_EXPECTATIONS) ? CPECAN_MODE_EXPECTATIONS : CPECAN_MODE_POSTERIOR,
                             CPECAN_KERNEL_GENERAL, flags & ~CPECAN_FLAG_EXPECTATIONS, out, true);
}

int cpecan_hip_batch_create_hdp(cpecan_ctx *c, const cpecan_item *items, int64_t nItems,
                                const char *xChars, int64_t nX, const double *events, int64_t nEvents,
                                const int64_t *anchors, int64_t nAnchorPairs,
                                const cpecan_band_params *params, int32_t flags, cpecan_batch **out) {
    if (!events) return fail(CPECAN_EINVAL, "bad argument");
    return batch_create_impl(c, items, nItems, xChars, nX, events, nullptr, nEvents, anchors, nAnchorPairs, params,
                             (flags & CPECAN_FLAG_EXPECTATIONS) ? CPECAN_MODE_EXPECTATIONS : CPECAN_MODE_POSTERIOR,
                             CPECAN_KERNEL_GENERAL, flags & ~CPECAN_FLAG_EXPECTATIONS, out, false, true);
}

int cpecan_hip_batch_create_dna(cpecan_ctx *c, const cpecan_item *items, int64_t nItems,
                                const char *xChars, int64_t nX, const char *yChars, int64_t nY,
                                const int64_t *anchors, int64_t nAnchorPairs,
                                const cpecan_band_params *params, int32_t flags, cpecan_batch **out) {
    if (!yChars) return fail(CPECAN_EINVAL, "bad argument");
    return batch_create_impl(c, items, nItems, xChars, nX, nullptr, yChars, nY, anchors, nAnchorPairs, params,
                             (flags & CPECAN_FLAG_EXPECTATIONS) ? CPECAN_MODE_EXPECTATIONS : CPECAN_MODE_POSTERIOR,
                             CPECAN_KERNEL_GENERAL, flags & ~CPECAN_FLAG_EXPECTATIONS, out);
}

int cpecan_hip_batch_create_sm4(cpecan_ctx *c, const cpecan_item *items, int64_t nItems, const char *xChars, int64_t nX,
                                const double *events, int64_t nEvents, const int64_t *anchors, int64_t nAnchorPairs,
                                const cpecan_band_params *params, int32_t flags, cpecan_batch **out) {
    if (flags & CPECAN_FLAG_EXPECTATIONS) return fail(CPECAN_EINVAL, "4-state batches: posterior decode only");
    return batch_create_impl(c, items, nItems, xChars, nX, events, nullptr, nEvents, anchors, nAnchorPairs, params,
                             CPECAN_MODE_POSTERIOR, CPECAN_KERNEL_GENERAL, flags, out, false, false, true);
}

int cpecan_hip_batch_run(cpecan_batch *b) { return cpecan_hip_batch_run_after(b, nullptr); }

int cpecan_hip_batch_run_after(cpecan_batch *b, cpecan_batch *after) {
    if (!b) return fail(CPECAN_EINVAL, "batch is NULL");
    cpecan_ctx *c = b->ctx;
    if (b->modelEpoch != c->modelEpoch)
        return fail(CPECAN_EINVAL, "cpecan_hip_models_clear was called on the context after this batch was created: "
                    "its model ids are gone");
    if (after && after->device != b->device) return fail(CPECAN_EINVAL, "the two batches live on different devices");
    HIP_TRY(hipSetDevice(c->device));
    if (after && after != b && after->ran) {
        /* behind the other batch on the device.  A batch of the wave kernels in one stream group is followed as soon as
         * its LAST FORWARD sweep is over: this batch's first forward sweep then shares the SIMDs with that batch's last
         * sweep back, as the forward sweep of that batch's own next window would have, and the totals, decode, counts
         * and packing of that batch run beside this one's first window. */
        hipEvent_t done = after->ev2;
        if (after->kernel == CPECAN_KERNEL_SYSTOLIC && after->sy && after->sy->wave && after->nGroups == 1 && after->nWindows > 0 &&
            after->evStage.size() == (size_t) (4 * after->nWindows + 1))
            done = after->evStage[(size_t) (1 + 4 * (after->nWindows - 1) + 1)];
        HIP_TRY(hipStreamWaitEvent(c->stream, done, 0));
    }
    b->countsValid = false;
    HIP_TRY(hipEventRecord(b->ev0, c->stream));
    if (b->mode == CPECAN_MODE_EXPECTATIONS)
        HIP_TRY(hipMemsetAsync(b->expect.p, 0, b->expect.n * sizeof(double), c->stream));
    HIP_TRY(hipEventRecord(b->ev1, c->stream));
    static const bool wave5Off = getenv("CPECAN_DNA_GENERAL") != nullptr; /* (tests, timing: the general kernel) */
    if (b->dna && !b->P.debug && !b->P.unbanded && b->maxWidth <= 192 && !wave5Off && !(b->flags & CPECAN_FLAG_GENERAL_KERNEL)) {
        /* the 5-state machine for bands a wave covers in one to three cells per lane: one wave per alignment, the
         * recurrence in registers (cpecan_kernel_wave5.hip), posterior decode or expectations */
        const bool em = b->mode == CPECAN_MODE_EXPECTATIONS;
        /* a batch that leaves SIMDs with fewer than two waves runs the sweeps of an alignment on a pair of waves
         * (forward and back overlapping); CPECAN_WAVE5_PAIRED=0/1 forces either form (tests, timing) */
        const char *pairedEnv = getenv("CPECAN_WAVE5_PAIRED");
        const int l5 = b->maxWidth <= 64 ? 0 : b->maxWidth <= 128 ? 1 : 2;
        /* (the one-wave E-step at three cells per lane needs more registers than two waves of a SIMD can have) */
        const bool paired = pairedEnv ? atoi(pairedEnv) != 0 : (b->nItems < CP_WAVE5_PAIRED_BELOW || (em && l5 == 2));
        static const auto kernels5 = std::array<decltype(&cpecan_k_wave5_l1), 12>{
            cpecan_k_wave5_l1, cpecan_k_wave5_l2, cpecan_k_wave5_l3, cpecan_k_wave5e_l1, cpecan_k_wave5e_l2, cpecan_k_wave5e_l3,
            cpecan_k_wave5p_l1, cpecan_k_wave5p_l2, cpecan_k_wave5p_l3, cpecan_k_wave5pe_l1, cpecan_k_wave5pe_l2, cpecan_k_wave5pe_l3 };
        auto kernel5 = kernels5[(size_t) ((paired ? 6 : 0) + (em ? 3 : 0) + l5)];
        hipLaunchKernelGGL(kernel5, dim3((unsigned) b->nItems), dim3(paired ? 128 : 64), 0, c->stream, (const DevItem *) b->items.p, b->P,
                           (const int *) b->bandL.p, (const int *) b->bandR.p, (const long long *) b->cellPrefix.p,
                           (const char *) b->chars.p, (const char *) b->charsY.p, (const double *) c->models5.p,
                           b->Fstore.p, b->pairs.p, b->pairLogp.p, b->nPairs.p, b->totXay.p, b->totVal.p, b->nTot.p,
                           em ? b->expect.p : nullptr);
        HIP_TRY(hipGetLastError());
    } else if (b->dna) {
        /* the forward sweep's two previous diagonals live in LDS where the widest band fits (3 diagonals of 5 states:
         * 120 bytes per cell of width); CPECAN_GENERAL_LDS=0 keeps them in HBM (timing, tests) */
        static const bool ldsOff = getenv("CPECAN_GENERAL_LDS") != nullptr && atoi(getenv("CPECAN_GENERAL_LDS")) == 0;
        DevParams P5 = b->P;
        P5.ldsWidth = (!ldsOff && b->maxWidth <= 248) ? b->maxWidth : 0;
        hipLaunchKernelGGL(cpecan_k_general5, dim3((unsigned) b->nItems), dim3(256), (size_t) P5.ldsWidth * 120, c->stream,
                           (const DevItem *) b->items.p, P5, (const int *) b->bandL.p,
                           (const int *) b->bandR.p, (const long long *) b->cellPrefix.p,
                           (const char *) b->chars.p, (const char *) b->charsY.p,
                           (const double *) c->models5.p, b->Fstore.p, b->Bstore.p, b->pairs.p,
                           b->pairLogp.p, b->nPairs.p, b->totXay.p, b->totVal.p, b->nTot.p,
                           (double *) nullptr, b->mode == CPECAN_MODE_EXPECTATIONS ? b->expect.p : nullptr);
        HIP_TRY(hipGetLastError());
    } else if (b->hdp && b->kernel == CPECAN_KERNEL_GENERAL) {
        hipLaunchKernelGGL(cpecan_k_generalh, dim3((unsigned) b->nItems), dim3(256), 0, c->stream,
                           (const DevItem *) b->items.p, b->P, (const int *) b->bandL.p,
                           (const int *) b->bandR.p, (const long long *) b->cellPrefix.p,
                           (const int *) b->kid.p, (const double *) b->events.p,
                           (const DevHdpModel *) c->modelsH.p, b->Fstore.p, b->Bstore.p, b->pairs.p,
                           b->pairLogp.p, b->nPairs.p, b->totXay.p, b->totVal.p, b->nTot.p,
                           b->mode == CPECAN_MODE_EXPECTATIONS ? b->expect.p : nullptr);
        HIP_TRY(hipGetLastError());
    } else if (b->sm4) {
        hipLaunchKernelGGL(cpecan_k_general4, dim3((unsigned) b->nItems), dim3(256), 0, c->stream,
                           (const DevItem *) b->items.p, b->P, (const int *) b->bandL.p, (const int *) b->bandR.p,
                           (const long long *) b->cellPrefix.p, (const unsigned short *) b->kidx.p,
                           (const double *) b->events.p, (const double *) c->models4.p, b->Fstore.p, b->Bstore.p, b->pairs.p,
                           b->pairLogp.p, b->nPairs.p, b->totXay.p, b->totVal.p, b->nTot.p);
        HIP_TRY(hipGetLastError());
    } else if (b->vanilla && b->kernel == CPECAN_KERNEL_GENERAL) {
        hipLaunchKernelGGL(cpecan_k_generalv, dim3((unsigned) b->nItems), dim3(256), 0, c->stream,
                           (const DevItem *) b->items.p, b->P, (const int *) b->bandL.p,
                           (const int *) b->bandR.p, (const long long *) b->cellPrefix.p,
                           (const unsigned short *) b->kidx.p, (const double *) b->events.p,
                           (const double *) b->logNoise.p, (const double *) c->modelsV.p, b->Fstore.p,
                           b->Bstore.p, b->pairs.p, b->pairLogp.p, b->nPairs.p, b->totXay.p, b->totVal.p,
                           b->nTot.p, b->mode == CPECAN_MODE_EXPECTATIONS ? b->expect.p : nullptr);
        HIP_TRY(hipGetLastError());
    } else if (b->kernel == CPECAN_KERNEL_GENERAL) {
        hipLaunchKernelGGL(cpecan_k_general, dim3((unsigned) b->nItems), dim3(256), 0, c->stream,
                           (const DevItem *) b->items.p, b->P, (const int *) b->bandL.p,
                           (const int *) b->bandR.p, (const long long *) b->cellPrefix.p,
                           (const unsigned short *) b->kidx.p, (const double *) b->events.p,
                           (const double *) c->models.p, b->Fstore.p, b->Bstore.p, b->pairs.p,
                           b->pairLogp.p, b->nPairs.p, b->totXay.p, b->totVal.p, b->nTot.p,
                           b->dbgB.p, b->mode == CPECAN_MODE_EXPECTATIONS ? b->expect.p : nullptr);
        HIP_TRY(hipGetLastError());
    } else {
        /* one pass: the per-item track of emission constants (a function of the inputs, rebuilt every run inside the
         * timed region), then for every traceback window the forward kernel and the backward kernel.  The wave
         * kernels run the two on streams of their own: the sweep back of window w overlaps the forward sweep of
         * window w+1 of the same alignments (each SIMD then holds a forward and a backward wave), and forward w+2,
         * which re-uses window w's ring rows, waits for the sweep back of w.  Events around every kernel give
         * per-kernel times and carry the dependencies. */
        const int G = b->nGroups, perGroup = 4 * b->nWindows + 1;
        if (b->evStage.size() != (size_t) (G * perGroup)) {
            for (hipEvent_t e : b->evStage) (void) hipEventDestroy(e);
            b->evStage.assign((size_t) (G * perGroup), nullptr);
            for (auto &e : b->evStage) HIP_TRY(hipEventCreate(&e));
        }
        /* does any model let gap Y switch to gap X?  (the nanopore default does not, stateMachine.c:1287: the
         * kernels then run the build without that term) */
        int withSwitch = 0;
        int rc;
        /* the models as the sweeps read them: strawMan tables, or the HDP records of an HDP batch */
        const double *models = b->hdp ? (const double *) c->modelsH.p : b->vanilla ? c->modelsV.p : c->models.p;
        if (b->vanilla) { /* (no gap Y -> gap X transition in this machine) */
            rc = cpecan_wave_launch_track_vanilla(c->stream, b->items.p, b->nItems, b->track.p, b->trackBase.p,
                                                  b->kidx.p, c->modelsV.p, b->syStates.p, b->maxLX);
        } else if (b->hdp) {
            for (const DevHdpModel &m : c->hostModelsH)
                if (m.t[T_GAP_SWITCH_TO_X] > -INFINITY) withSwitch = 1;
            rc = cpecan_wave_launch_track_hdp(c->stream, b->items.p, b->nItems, b->track.p, b->trackBase.p, b->kid.p,
                                              c->modelsH.p, b->syStates.p, b->maxLX);
        } else {
            for (int m = 0; m < c->nModels; m++)
                if (c->switchToX[(size_t) m] > -INFINITY) withSwitch = 1;
            rc = (b->sy->wave ? cpecan_wave_launch_track : cpecan_systolic_launch_track)(
                c->stream, b->items.p, b->nItems, b->track.p, b->trackBase.p, b->kidx.p, c->models.p, b->syStates.p,
                b->maxLX);
        }
        /* the assembly sweeps (no model of the batch may let gap Y switch to gap X: they have no such term) */
        const bool asmRun = b->useAsm && !withSwitch && rc == 0;
        AsmArgs asmArgs{};
        if (asmRun) {
            asmArgs.items = b->items.p; asmArgs.trackBase = b->trackBase.p; asmArgs.planWin = b->planWin.p;
            asmArgs.planCtl = b->planCtl.p; asmArgs.planOff = b->planOff.p; asmArgs.events = b->events.p;
            asmArgs.models = c->models.p; asmArgs.track = b->track.p; asmArgs.ring = b->Fstore.p;
            asmArgs.ringDoubles = b->ringDoubles; asmArgs.states = b->syStates.p; asmArgs.ctx = b->asmCtx.p;
            asmArgs.ctxBytes = ASM_CTX_BYTES; asmArgs.coef = cpecan_asm_coef(c->device); asmArgs.nItems = (int) b->nItems;
            asmArgs.ringD = b->ringD; asmArgs.maxWindows = b->asmMaxWindows; asmArgs.scratch = b->syScratch.p;
            asmArgs.scratchBytes = b->scratchBytes; asmArgs.logThrSlack = b->P.logThrSlack; asmArgs.modelStride = CP_MODEL_STRIDE;
            asmArgs.maskTab = b->asmMasks.p;
            rc = cpecan_asm_launch_begin(c->stream, b->items.p, b->nItems, b->Fstore.p, b->ringDoubles);
            if (getenv("CPECAN_ASM_TRACE")) {
                auto span = [](const char *what, const void *p, size_t bytes) {
                    fprintf(stderr, "[cpecan asm]   %-10s %p .. %p (%zu bytes)\n", what, p, (const char *) p + bytes, bytes);
                };
                span("items", b->items.p, b->items.n * sizeof(DevItem));
                span("trackBase", b->trackBase.p, b->trackBase.n * 8);
                span("planWin", b->planWin.p, b->planWin.n * sizeof(AsmPlanWin));
                span("planCtl", b->planCtl.p, b->planCtl.n * sizeof(AsmPlanCtl));
                span("planOff", b->planOff.p, b->planOff.n * 8);
                span("events", b->events.p, b->events.n * 8);
                span("models", c->models.p, c->models.n * 8);
                span("track", b->track.p, b->track.n * 8);
                span("ring", b->Fstore.p, b->Fstore.n * 8);
                span("states", b->syStates.p, b->syStates.n);
                span("ctx", b->asmCtx.p, b->asmCtx.n);
                span("scratch", b->syScratch.p, b->syScratch.n);
                span("masks", b->asmMasks.p, b->asmMasks.n * 4);
                span("coef", asmArgs.coef, 512);
                fprintf(stderr, "[cpecan asm]   ringD %d ringDoubles %lld windows %d scratchBytes %lld\n", b->ringD, b->ringDoubles,
                        b->asmMaxWindows, b->scratchBytes);
            }
        }
        HIP_TRY(hipEventRecord(b->evFork, c->stream));
        const long long per = (b->nItems + G - 1) / G;
        for (int gi = 0; gi < G && rc == 0; gi++) {
            const long long i0 = gi * per, n = std::min<long long>(per, b->nItems - i0);
            hipStream_t sF = b->gStream[(size_t) gi], sB = b->sy->wave ? b->gStreamB[(size_t) gi] : sF;
#ifdef CPECAN_TIMING_BUILD
            if (getenv("CPECAN_TIMING_SERIAL")) sB = sF; /* timing study: every sweep alone on the chip */
#endif
            hipEvent_t *ev = b->evStage.data() + (size_t) gi * perGroup;
            HIP_TRY(hipStreamWaitEvent(sF, b->evFork, 0));
            if (sB != sF) HIP_TRY(hipStreamWaitEvent(sB, b->evFork, 0));
            HIP_TRY(hipEventRecord(ev[0], sF));
            const long long bringRow = b->sy->bring_row_doubles();
            for (int w = 0; w < b->nWindows && rc == 0; w++) {
                hipEvent_t *e4 = ev + 1 + 4 * w;
                /* assembly sweeps: the window's totals and decode (and the re-sweep of what cannot be trusted) run on a
                 * stream of their own.  The sweep back of the next window does not wait for them (it fills the other
                 * half of the scratch), nor does the forward sweep of window w+2 (the ring holds three windows, the
                 * state four window records); what does: the sweep back of w+2 (scratch), the forward sweep of w+3
                 * (ring rows, window record).  The forward sweep of w+2 still waits for the sweep back of w, which reads
                 * the context that sweep will overwrite when it ends. */
                const bool postAside = asmRun && b->asmBackward && sB != sF && b->asmPost != nullptr;
                char *scratchW = b->syScratch.p + (postAside ? (size_t) (w & 1) * (size_t) b->nItems * (size_t) b->scratchBytes : 0);
                if (sB != sF && w >= 2) HIP_TRY(hipStreamWaitEvent(sF, ev[1 + 4 * (w - 2) + 3], 0));
                if (postAside && w >= 3) HIP_TRY(hipStreamWaitEvent(sF, b->evPost[(size_t) w - 3], 0));
                HIP_TRY(hipEventRecord(e4[0], sF));
                /* the kernels index everything per alignment by blockIdx: shift the bases */
                if (n > 0 && asmRun) {
                    asmArgs.window = w;
                    rc = cpecan_asm_launch_forward(c->device, sF, &asmArgs);
                } else if (n > 0)
                    rc = b->sy->launch_forward(sF, b->items.p + i0, n, b->P, b->bandTab.p, b->track.p,
                                               b->trackBase.p + i0, b->events.p, models,
                                               b->Fstore.p + i0 * b->ringDoubles, b->ringDoubles, b->ringD,
                                               b->syStates.p + i0 * b->stateBytes, w, withSwitch);
                HIP_TRY(hipEventRecord(e4[1], sF));
                if (sB != sF) HIP_TRY(hipStreamWaitEvent(sB, e4[1], 0));
                HIP_TRY(hipEventRecord(e4[2], sB));
#ifdef CPECAN_TIMING_BUILD
                static const bool fwdOnly = getenv("CPECAN_TIMING_FORWARD_ONLY") != nullptr; /* timing study: wrong results */
#else
                const bool fwdOnly = false;
#endif
                if (rc == 0 && n > 0 && asmRun && b->asmBackward && !fwdOnly) {
                    asmArgs.window = w;
                    asmArgs.scratch = scratchW;
                    if (postAside && w >= 2) HIP_TRY(hipStreamWaitEvent(sB, b->evPost[(size_t) w - 2], 0));
                    rc = cpecan_asm_launch_backward(c->device, sB, &asmArgs);
                    hipStream_t sP = postAside ? b->asmPost : sB;
                    if (postAside) {
                        HIP_TRY(hipEventRecord(e4[3], sB));
                        HIP_TRY(hipStreamWaitEvent(sP, e4[3], 0));
                    }
#ifdef CPECAN_TIMING_BUILD
                    static const bool noPost = getenv("CPECAN_TIMING_NO_POST") != nullptr; /* timing study: sweeps only */
#else
                    const bool noPost = false;
#endif
                    if (rc == 0 && !noPost)
                        rc = cpecan_wave_launch_post_asm_l3(sP, b->items.p, n, b->P, b->bandTab.p, b->track.p, b->trackBase.p, models,
                                                            b->Fstore.p, b->ringDoubles, b->ringD, b->syStates.p, b->pairs.p,
                                                            b->pairLogp.p, b->totXay.p, b->totVal.p, scratchW,
                                                            b->scratchBytes, w);
                    if (postAside) HIP_TRY(hipEventRecord(b->evPost[(size_t) w], sP));
                } else if (rc == 0 && n > 0 && !fwdOnly)
                    rc = b->sy->launch_backward(sB, b->items.p + i0, n, b->P, b->bandTab.p, b->track.p,
                                                b->trackBase.p + i0, models,
                                                b->Fstore.p + i0 * b->ringDoubles, b->ringDoubles, b->ringD,
                                                b->syStates.p + i0 * b->stateBytes, b->pairs.p, b->pairLogp.p,
                                                b->totXay.p, b->totVal.p, b->syScratch.p + i0 * b->scratchBytes,
                                                b->scratchBytes,
                                                b->Bring.p ? b->Bring.p + i0 * (long long) b->ringD * bringRow : nullptr,
                                                w, withSwitch);
                if (rc == 0 && n > 0 && b->mode == CPECAN_MODE_EXPECTATIONS)
                    rc = b->sy->launch_expect(sB, b->items.p + i0, n, b->P, b->bandTab.p, b->track.p,
                                              b->trackBase.p + i0, b->kidx.p, models,
                                              b->Fstore.p + i0 * b->ringDoubles, b->ringDoubles,
                                              b->Bring.p + i0 * (long long) b->ringD * bringRow, b->ringD,
                                              b->syStates.p + i0 * b->stateBytes,
                                              b->syScratch.p + i0 * b->scratchBytes, b->scratchBytes, b->expect.p, w,
                                              b->pairs.p, b->pairLogp.p);
                if (!postAside) HIP_TRY(hipEventRecord(e4[3], sB));
            }
            if (asmRun && b->asmBackward && sB != sF && b->asmPost != nullptr && b->nWindows > 0) /* the last post kernel follows all */
                HIP_TRY(hipStreamWaitEvent(sB, b->evPost[(size_t) b->nWindows - 1], 0));
            HIP_TRY(hipEventRecord(b->evJoin[(size_t) gi], sB)); /* the last sweep back follows every forward sweep */
            HIP_TRY(hipStreamWaitEvent(c->stream, b->evJoin[(size_t) gi], 0));
        }
        if (rc == 0)
            rc = (b->sy->wave ? cpecan_wave_launch_counts : cpecan_systolic_launch_counts)(
                c->stream, b->syStates.p, b->nItems, b->nPairs.p, b->nTot.p, b->nCells.p);
        if (rc != 0) return fail(CPECAN_EHIP, "throughput kernel launch failed: %s",
                                 hipGetErrorString(hipGetLastError()));
    }
    /* posterior decode: the run ends with its candidates packed for the host (16-byte records + the device's verdict),
     * into a buffer sized by a guess the first time (about one candidate per diagonal) and by the last run's count
     * afterwards; ensure_counts packs again if the guess was short.  Done here, inside the pass, because a kernel
     * launched later would wait for wave slots behind the next batch's sweeps. */
    b->packedInRun = false;
    static const bool packInRun = getenv("CPECAN_PACK_LATER") == nullptr;
    if (packInRun && b->mode == CPECAN_MODE_POSTERIOR && !b->P.debug) {
        if (b->packed.n == 0) {
            long long guess = 0;
            for (const DevItem &d : b->hItems) guess += std::min<long long>(d.pairCap, d.lX + d.lY + 64);
            HIP_TRY(b->packed.alloc((size_t) guess));
            HIP_TRY(b->packedPost.alloc((size_t) guess));
        }
        if (b->packBase.n < (size_t) b->nItems + 1) HIP_TRY(b->packBase.alloc((size_t) b->nItems + 1));
        if (b->undecided.n == 0) HIP_TRY(b->undecided.alloc(1 + 2 * CP_UNDECIDED_CAP));
        HIP_TRY(hipMemsetAsync(b->undecided.p, 0, sizeof(long long), c->stream));
        hipLaunchKernelGGL(cpecan_k_pack_base, dim3(1), dim3(256), 0, c->stream, (const DevItem *) b->items.p,
                           (const long long *) b->nPairs.p, (long long) b->nItems, b->packBase.p);
        hipLaunchKernelGGL(cpecan_k_pack_pairs, dim3((unsigned) b->nItems), dim3(256), 0, c->stream,
                           (const DevItem *) b->items.p, (const long long *) b->packBase.p, (const long long *) b->pairs.p,
                           (const double *) b->pairLogp.p, b->P.threshold, (long long) b->packed.n, b->packed.p,
                           b->packedPost.p, b->undecided.p, b->compactPairs ? 1 : 0);
        HIP_TRY(hipGetLastError());
        b->packedInRun = true;
    }
    HIP_TRY(hipEventRecord(b->ev2, c->stream));
    b->ran = true;
    return CPECAN_OK;
}

int cpecan_hip_batch_systolic_rows(cpecan_batch *b, int32_t *rows) {
    if (!b || !rows) return fail(CPECAN_EINVAL, "bad argument");
    if (b->kernel != CPECAN_KERNEL_SYSTOLIC) return fail(CPECAN_EINVAL, "not a systolic batch");
    *rows = b->sy->rows;
    return CPECAN_OK;
}

int cpecan_hip_batch_kernel_family(cpecan_batch *b, int32_t *wave) {
    if (!b || !wave) return fail(CPECAN_EINVAL, "bad argument");
    if (b->dna) { /* the 5-state machine: one wave per alignment (cpecan_kernel_wave5.hip) where batch_run picks it */
        static const bool wave5Off = getenv("CPECAN_DNA_GENERAL") != nullptr;
        *wave = (!b->P.debug && !b->P.unbanded && b->maxWidth <= 192 && !wave5Off && !(b->flags & CPECAN_FLAG_GENERAL_KERNEL)) ? 1 : 0;
        return CPECAN_OK;
    }
    if (b->kernel != CPECAN_KERNEL_SYSTOLIC) return fail(CPECAN_EINVAL, "not a register-resident batch");
    *wave = b->sy->wave ? 1 : 0;
    return CPECAN_OK;
}

int cpecan_hip_batch_assembly_sweeps(cpecan_batch *b, int32_t *sweeps) {
    if (!b || !sweeps) return fail(CPECAN_EINVAL, "bad argument");
    *sweeps = b->useAsm ? (b->asmBackward ? 2 : 1) : 0;
    return CPECAN_OK;
}

int cpecan_hip_batch_stage_ms(cpecan_batch *b, float *msForward, float *msBackward, int32_t *launchesEach) {
    if (!b || !b->ran) return fail(CPECAN_EINVAL, "batch has not run");
    if (b->kernel != CPECAN_KERNEL_SYSTOLIC) return fail(CPECAN_EINVAL, "only the systolic path has stages");
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipEventSynchronize(b->ev2));
    float f = 0, k = 0;
    const int perGroup = 4 * b->nWindows + 1;
    for (int gi = 0; gi < b->nGroups; gi++)
        for (int w = 0; w < b->nWindows; w++) {
            const hipEvent_t *ev = b->evStage.data() + (size_t) gi * perGroup + 1 + 4 * w;
            float a = 0, c2 = 0;
            HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
            HIP_TRY(hipEventElapsedTime(&c2, ev[2], ev[3]));
            f += a;
            k += c2;
        }
    static const bool timeline = getenv("CPECAN_TIMELINE") != nullptr; /* timing study: when every stage began and ended */
    if (timeline) {
        const hipEvent_t *ev = b->evStage.data();
        for (int w = 0; w < b->nWindows; w++) {
            float t[4] = { 0, 0, 0, 0 };
            for (int q = 0; q < 4; q++) (void) hipEventElapsedTime(&t[q], ev[0], ev[1 + 4 * w + q]);
            fprintf(stderr, "[cpecan timeline] window %2d: forward %7.3f .. %7.3f   backward+post %7.3f .. %7.3f\n", w, t[0], t[1],
                    t[2], t[3]);
        }
    }
    if (msForward) *msForward = f;
    if (msBackward) *msBackward = k;
    if (launchesEach) *launchesEach = b->nWindows * b->nGroups;
    return CPECAN_OK;
}

int cpecan_hip_batch_shader_clock_mhz(cpecan_batch *b, double *mhz) {
    if (!b || !mhz) return fail(CPECAN_EINVAL, "bad argument");
    *mhz = 0.0;
    if (!b->ran || b->kernel != CPECAN_KERNEL_SYSTOLIC || !b->sy->wave) return CPECAN_OK; /* not measured on this path */
    HIP_TRY(hipSetDevice(b->ctx->device));
    if (cpecan_wave_shader_clock_mhz(b->ctx->stream, b->syStates.p, b->nItems, mhz) != 0)
        return fail(CPECAN_EHIP, "reading the sweeps' clock counters failed");
    return CPECAN_OK;
}

int cpecan_hip_batch_info(cpecan_batch *b, int32_t *kernel, int32_t *workgroups, int32_t *maxWidth) {
    if (!b) return fail(CPECAN_EINVAL, "batch is NULL");
    if (kernel) *kernel = b->kernel;
    if (workgroups) *workgroups = b->kernel == CPECAN_KERNEL_GENERAL ? (int32_t) b->nItems : b->nWorkers;
    if (maxWidth) *maxWidth = b->maxWidth;
    return CPECAN_OK;
}

int cpecan_hip_batch_sync(cpecan_batch *b) {
    if (!b) return fail(CPECAN_EINVAL, "batch is NULL");
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    return CPECAN_OK;
}

int cpecan_hip_batch_elapsed_ms(cpecan_batch *b, float *msTotal, float *msKernel) {
    if (!b || !b->ran) return fail(CPECAN_EINVAL, "batch has not run");
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipEventSynchronize(b->ev2));
    float a = 0, k = 0;
    HIP_TRY(hipEventElapsedTime(&a, b->ev0, b->ev2));
    HIP_TRY(hipEventElapsedTime(&k, b->ev1, b->ev2));
    if (msTotal) *msTotal = a;
    if (msKernel) *msKernel = k;
    return CPECAN_OK;
}

/* Counts and aligned pairs of a finished run.  If an alignment produced more pairs than its share of the pair buffer
 * holds (flat posteriors: a tiny threshold, the HDP machine's linear densities), the buffer is re-laid-out to the
 * reported counts and the batch is run once more -- the reference returns the list whatever its length.  The device
 * selects pairs by the exponent (F+B)-total with a margin below log(threshold); exp(), the exact threshold test and
 * floor(p * 1e7) are finished here with the host libm, the one the reference calls
 * (diagonalCalculationPosteriorMatchProbs, impl/pairwiseAligner.c:776-786). */
/* item i's first packBase[i + 1] - packBase[i] candidates, from its own region of the pair buffers to the packed one */
/* With every candidate goes a verdict on its integer posterior: the device's exp() and the host libm's differ by at
 * most a few units in the last place, so wherever exp(logp) is not within a (far wider) margin of the threshold, of 1
 * or of a multiple of 1e-7, floor(p * 1e7) is the same number on both and is taken here (post >= 0), or the pair is
 * surely below the threshold (post -2); the few that are close (post -1) are finished by the host with its libm. */
extern "C" __global__ void cpecan_k_pack_pairs(const DevItem *items, const long long *packBase, const long long *pairs,
                                               const double *logp, double threshold, long long capacity,
                                               PackedPair *out, int *post, long long *undecided /* [0] count, then
                                               CP_UNDECIDED_CAP x (packed index, exponent bits) */,
                                               int compact /* both coordinates below 65536: four bytes a pair */) {
    const DevItem &d = items[blockIdx.x];
    if (packBase[gridDim.x] > capacity) return; /* (packed at the end of a run into a buffer sized by a guess: the host
                                                   sees the same total and packs again into one that fits) */
    const long long o = packBase[blockIdx.x], n = packBase[blockIdx.x + 1] - o;
    for (long long k = threadIdx.x; k < n; k += blockDim.x) {
        PackedPair r;
        r.x = (int) pairs[(d.pairBase + k) * 3 + 1];
        r.y = (int) pairs[(d.pairBase + k) * 3 + 2];
        if (compact) ((unsigned *) out)[o + k] = (unsigned) r.x | ((unsigned) r.y << 16);
        else out[o + k] = r;
        const double e = logp[d.pairBase + k];
        const double p = exp(e);
        int v = -1;
        if (p == p) {
            if (p < threshold - (1e-9 * threshold + 1e-300)) v = -2;
            else if (p > threshold + (1e-9 * threshold + 1e-300) || threshold == 0.0) {
                if (p > 1.0 + 1e-9) v = 10000000;
                else if (p < 1.0 - 1e-9) {
                    const double q = p * 10000000.0, fl = floor(q);
                    if (q - fl > 1e-5 && fl + 1.0 - q > 1e-5) v = (int) fl;
                } else if (e >= 0.0) v = 10000000; /* exp(e) >= 1 on any libm: clamped to 1 */
                else if (e <= -1e-15) v = 9999999; /* exp(e) <= 1 - 9e-16 < 1, and p * 1e7 rounds below 1e7 (its
                                                      ulp there is 1.9e-9, the deficit at least 1e-8): a quarter of a
                                                      C3 batch's candidates are this sure a match */
            }
        }
        post[o + k] = v;
        if (v == -1) { /* the host settles it: its exponent goes along (a short list; a batch that overflows it has
                          the host fetch the exponents item by item) */
            const unsigned long long j = atomicAdd((unsigned long long *) undecided, 1ull);
            if (j < CP_UNDECIDED_CAP) {
                undecided[1 + 2 * j] = o + k;
                undecided[2 + 2 * j] = __double_as_longlong(e);
            }
        }
    }
}

/* packBase[i] = candidates of the items before i (each item's count capped at its capacity), packBase[n] = all of
 * them: the offsets cpecan_k_pack_pairs writes to, formed on the device so that a run can end with its candidates
 * packed (the host forms the same sums from the counts it fetches) */
extern "C" __global__ __launch_bounds__(256) void cpecan_k_pack_base(const DevItem *items, const long long *nPairs,
                                                                     long long nItems, long long *packBase) {
    __shared__ long long part[256];
    const long long per = (nItems + 255) / 256, i0 = threadIdx.x * per, i1 = i0 + per < nItems ? i0 + per : nItems;
    long long sum = 0;
    for (long long i = i0; i < i1; i++) sum += nPairs[i] < items[i].pairCap ? nPairs[i] : items[i].pairCap;
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int t = 0; t < 256; t++) {
            const long long v = part[t];
            part[t] = run;
            run += v;
        }
        packBase[nItems] = run;
    }
    __syncthreads();
    long long run = part[threadIdx.x];
    for (long long i = i0; i < i1; i++) {
        packBase[i] = run;
        run += nPairs[i] < items[i].pairCap ? nPairs[i] : items[i].pairCap;
    }
}

static int ensure_counts(cpecan_batch *b) {
    if (!b->ran) return fail(CPECAN_EINVAL, "batch has not run");
    if (b->countsValid) return CPECAN_OK;
    HIP_TRY(hipSetDevice(b->ctx->device));
    Lap lap("ensure_counts");
    b->hNPairs.resize((size_t) b->nItems);
    b->hNTot.resize((size_t) b->nItems);
    for (int attempt = 0;; attempt++) {
        HIP_TRY(hipStreamSynchronize(b->ctx->stream));
        HIP_TRY(hipMemcpy(b->hNPairs.data(), b->nPairs.p, (size_t) b->nItems * sizeof(long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(b->hNTot.data(), b->nTot.p, (size_t) b->nItems * sizeof(long long), hipMemcpyDeviceToHost));
        bool over = false;
        for (int64_t i = 0; i < b->nItems; i++)
            if (b->hNPairs[(size_t) i] > b->hItems[(size_t) i].pairCap) over = true;
        if (!over) break;
        if (attempt == 2) return fail(CPECAN_EOVERFLOW, "aligned-pair counts keep growing between identical runs");
        long long total = 0;
        for (int64_t i = 0; i < b->nItems; i++) {
            DevItem &d = b->hItems[(size_t) i];
            d.pairCap = std::max(d.pairCap, b->hNPairs[(size_t) i] + 64);
            d.pairBase = total;
            total += d.pairCap;
        }
        HIP_TRY(b->pairs.alloc((size_t) total * 3));
        HIP_TRY(b->pairLogp.alloc((size_t) total));
        HIP_TRY(hipMemcpy(b->items.p, b->hItems.data(), (size_t) b->nItems * sizeof(DevItem), hipMemcpyHostToDevice));
        int rc = cpecan_hip_batch_run(b);
        if (rc != CPECAN_OK) return rc;
    }
    b->hPairBase.assign((size_t) b->nItems + 1, 0);
    for (int64_t i = 0; i < b->nItems; i++)
        b->hPairBase[(size_t) i + 1] = b->hPairBase[(size_t) i] + std::min(b->hNPairs[(size_t) i], b->hItems[(size_t) i].pairCap);
    const long long all = b->hPairBase[(size_t) b->nItems];
    if (b->mode != CPECAN_MODE_POSTERIOR) {
        b->hPairs.resize((size_t) all * 3);
        b->hLogp.resize((size_t) all);
        /* (in expectation mode the HDP machine's pair buffer carries event-to-k-mer assignments, not posteriors:
         * short lists, copied as they are) */
        for (int64_t i = 0; i < b->nItems; i++) {
            const DevItem &d = b->hItems[(size_t) i];
            const long long n = b->hPairBase[(size_t) i + 1] - b->hPairBase[(size_t) i], o = b->hPairBase[(size_t) i];
            if (n == 0) continue;
            HIP_TRY(hipMemcpyAsync(b->hPairs.data() + o * 3, b->pairs.p + d.pairBase * 3, (size_t) n * 3 * sizeof(long long),
                                   hipMemcpyDeviceToHost, b->ctx->stream));
            HIP_TRY(hipMemcpyAsync(b->hLogp.data() + o, b->pairLogp.p + d.pairBase, (size_t) n * sizeof(double),
                                   hipMemcpyDeviceToHost, b->ctx->stream));
        }
        HIP_TRY(hipStreamSynchronize(b->ctx->stream));
        if (b->hdp && b->kernel == CPECAN_KERNEL_SYSTOLIC) {
            /* the HDP machine's event assignments from the wave kernels: appended by whichever thread got there, each
             * tagged with its traceback window (first field = from-state + 4 * window).  The reference walks windows
             * upwards, inside a window the diagonals downwards, a diagonal by ascending x, a cell by from-state
             * (cell_signal_updateTransAndKmerSkipExpectations2 inside diagonalCalculation_Expectations): put them so */
            std::vector<long long> order;
            std::vector<long long> tri;
            std::vector<double> lp;
            for (int64_t i = 0; i < b->nItems; i++) {
                const long long o = b->hPairBase[(size_t) i], n = b->hPairBase[(size_t) i + 1] - o;
                if (n <= 1) {
                    if (n == 1) b->hPairs[(size_t) o * 3] &= 3;
                    continue;
                }
                long long *p3 = b->hPairs.data() + o * 3;
                double *pl = b->hLogp.data() + o;
                order.resize((size_t) n);
                for (long long k = 0; k < n; k++) order[(size_t) k] = k;
                std::sort(order.begin(), order.end(), [p3](long long a, long long c2) {
                    const long long wa = p3[3 * a] >> 2, wc = p3[3 * c2] >> 2;
                    if (wa != wc) return wa < wc;
                    const long long da = p3[3 * a + 1] + p3[3 * a + 2], dc = p3[3 * c2 + 1] + p3[3 * c2 + 2];
                    if (da != dc) return da > dc;
                    if (p3[3 * a + 1] != p3[3 * c2 + 1]) return p3[3 * a + 1] < p3[3 * c2 + 1];
                    return (p3[3 * a] & 3) < (p3[3 * c2] & 3);
                });
                tri.assign(p3, p3 + 3 * n);
                lp.assign(pl, pl + n);
                for (long long k = 0; k < n; k++) {
                    const long long src = order[(size_t) k];
                    p3[3 * k] = tri[(size_t) (3 * src)] & 3;
                    p3[3 * k + 1] = tri[(size_t) (3 * src + 1)];
                    p3[3 * k + 2] = tri[(size_t) (3 * src + 2)];
                    pl[k] = lp[(size_t) src];
                }
            }
        }
        b->countsValid = true;
        return CPECAN_OK;
    }
    if (all > 0) {
        const bool packedAlready = b->packedInRun && (size_t) all <= b->packed.n; /* the run ended with them packed */
        if (b->packBase.n < (size_t) b->nItems + 1) HIP_TRY(b->packBase.alloc((size_t) b->nItems + 1));
        if (b->packed.n < (size_t) all) {
            HIP_TRY(b->packed.alloc((size_t) all + (size_t) all / 8));
            HIP_TRY(b->packedPost.alloc((size_t) all + (size_t) all / 8));
        }
        if (b->hPackedCap < (size_t) all) {
            if (b->hPacked) pinned_cache().put(b->hPacked, b->hPackedBlock);
            if (b->hPost) pinned_cache().put(b->hPost, b->hPostBlock);
            b->hPacked = nullptr;
            b->hPost = nullptr;
            b->hPackedCap = (size_t) all + (size_t) all / 8;
            HIP_TRY(pinned_cache().get((void **) &b->hPacked, b->hPackedCap * sizeof(PackedPair), &b->hPackedBlock));
            HIP_TRY(pinned_cache().get((void **) &b->hPost, b->hPackedCap * sizeof(int), &b->hPostBlock));
        }
        if (b->undecided.n == 0) HIP_TRY(b->undecided.alloc(1 + 2 * CP_UNDECIDED_CAP));
        if (!b->hUndecided)
            HIP_TRY(pinned_cache().get((void **) &b->hUndecided, (1 + 2 * CP_UNDECIDED_CAP) * sizeof(long long), &b->hUndecidedBlock));
        if (!packedAlready) {
        HIP_TRY(hipMemcpyAsync(b->packBase.p, b->hPairBase.data(), ((size_t) b->nItems + 1) * sizeof(long long),
                               hipMemcpyHostToDevice, b->ctx->stream));
        HIP_TRY(hipMemsetAsync(b->undecided.p, 0, sizeof(long long), b->ctx->stream));
        hipLaunchKernelGGL(cpecan_k_pack_pairs, dim3((unsigned) b->nItems), dim3(256), 0, b->ctx->stream,
                           (const DevItem *) b->items.p, (const long long *) b->packBase.p, (const long long *) b->pairs.p,
                           (const double *) b->pairLogp.p, b->P.threshold, (long long) b->packed.n, b->packed.p,
                           b->packedPost.p, b->undecided.p, b->compactPairs ? 1 : 0);
        HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipMemcpyAsync(b->hUndecided, b->undecided.p, (1 + 2 * CP_UNDECIDED_CAP) * sizeof(long long),
                               hipMemcpyDeviceToHost, b->ctx->stream));
        HIP_TRY(hipMemcpyAsync(b->hPacked, b->packed.p, (size_t) all * (b->compactPairs ? sizeof(unsigned) : sizeof(PackedPair)),
                               hipMemcpyDeviceToHost, b->ctx->stream));
        HIP_TRY(hipMemcpyAsync(b->hPost, b->packedPost.p, (size_t) all * sizeof(int), hipMemcpyDeviceToHost,
                               b->ctx->stream));
        HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    }
    /* The close calls: exp(), the threshold test and floor(p * 1e7) with the host's libm, the one the reference calls
     * (impl/pairwiseAligner.c:776-786), written back over the device's "undecided" verdict; and the number of pairs
     * every item keeps.  The records stay packed in pinned memory; cpecan_hip_batch_fetch_pairs expands an item's
     * pairs into the reference's triples when they are asked for.  Items are independent: dealt to the host threads. */
    const double threshold = b->P.threshold;
    int *verdict = b->hPost;
    static const bool hostOnly = getenv("CPECAN_HOST_FINALISE") != nullptr; /* (tests: every pair through the host libm) */
    const int nt = (int) std::min<int64_t>(all > 200000 ? host_threads() : 1, b->nItems);
    /* contiguous runs of items with about the same number of candidates each */
    std::vector<int64_t> cut(1, 0);
    for (int t = 0; t < nt; t++) {
        const long long want = all * (t + 1) / nt;
        int64_t i1 = cut.back();
        while (i1 < b->nItems && (b->hPairBase[(size_t) i1 + 1] <= want || t == nt - 1)) i1++;
        cut.push_back(i1);
    }
    cut.back() = b->nItems;
    /* what the device settled is counted by the host threads; what it left open (verdict -1) comes with its exponent
     * in the short list the pack kernel made, and is settled here with the host libm */
    const long long listed = all > 0 ? b->hUndecided[0] : 0;
    const bool byList = !hostOnly && listed <= (long long) CP_UNDECIDED_CAP;
    if (getenv("CPECAN_TIMING")) fprintf(stderr, "[cpecan timing] ensure_counts: %lld candidates, %lld left to the host\n", all, listed);
    auto settle = [threshold](double e) {
        double p = exp(e);
        if (!(p >= threshold)) return -2;
        if (p > 1.0) p = 1.0;
        return (int) floor(p * 10000000.0);
    };
    if (byList)
        for (long long j = 0; j < listed; j++) {
            double e;
            memcpy(&e, &b->hUndecided[2 + 2 * j], sizeof e);
            verdict[b->hUndecided[1 + 2 * j]] = settle(e);
        }
    std::atomic<int> failed{0};
    auto scan = [b, verdict, byList, &settle, &failed](int64_t i0, int64_t i1) {
        std::vector<double> e;
        for (int64_t i = i0; i < i1; i++) {
            const long long o = b->hPairBase[(size_t) i], n = b->hPairBase[(size_t) i + 1] - o;
            if (!byList && n > 0) { /* (tests, or more close calls than the list holds: this item's exponents from HBM) */
                e.resize((size_t) n);
                if (hipSetDevice(b->ctx->device) != hipSuccess ||
                    hipMemcpy(e.data(), b->pairLogp.p + b->hItems[(size_t) i].pairBase, (size_t) n * sizeof(double),
                              hipMemcpyDeviceToHost) != hipSuccess) {
                    failed = 1;
                    return;
                }
            }
            long long kept = 0;
            for (long long k = 0; k < n; k++) {
                if (!byList && (hostOnly || verdict[o + k] == -1)) verdict[o + k] = settle(e[(size_t) k]);
                kept += verdict[o + k] >= 0;
            }
            b->hNPairs[(size_t) i] = kept;
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++)
            if (cut[(size_t) t + 1] > cut[(size_t) t]) pool.emplace_back(scan, cut[(size_t) t], cut[(size_t) t + 1]);
        scan(cut[0], cut[1]);
        for (std::thread &th : pool) th.join();
    }
    if (failed) return fail(CPECAN_EHIP, "fetching the candidates' exponents failed: %s", hipGetErrorString(hipGetLastError()));
    b->countsValid = true;
    return CPECAN_OK;
}

int cpecan_hip_batch_counts(cpecan_batch *b, int64_t *nPairs, int64_t *nTotals, int64_t *nCells) {
    if (!b) return fail(CPECAN_EINVAL, "batch is NULL");
    int rc = ensure_counts(b);
    if (rc) return rc;
    for (int64_t i = 0; i < b->nItems; i++) {
        if (nPairs) nPairs[i] = b->hNPairs[(size_t) i];
        if (nTotals) nTotals[i] = b->hNTot[(size_t) i];
        if (nCells) nCells[i] = b->hNCells[(size_t) i];
    }
    return CPECAN_OK;
}

int cpecan_hip_batch_fetch_pairs(cpecan_batch *b, int64_t item, int64_t *triples, double *logp,
                                 int64_t cap) {
    if (!b || item < 0 || item >= b->nItems || !triples) return fail(CPECAN_EINVAL, "bad argument");
    int rc = ensure_counts(b);
    if (rc) return rc;
    const long long n = b->hNPairs[(size_t) item], o = b->hPairBase[(size_t) item];
    if (n > cap) return fail(CPECAN_EOVERFLOW, "need room for %lld triples", n);
    if (n == 0) return CPECAN_OK;
    if (b->mode != CPECAN_MODE_POSTERIOR) {
        memcpy(triples, b->hPairs.data() + o * 3, (size_t) n * 3 * sizeof(long long));
        if (logp) memcpy(logp, b->hLogp.data() + o, (size_t) n * sizeof(double));
        return CPECAN_OK;
    }
    /* the item's packed candidates with their settled verdicts -> (floor(p * 1e7), x, y), emission order */
    const long long cand = b->hPairBase[(size_t) item + 1] - o;
    std::vector<double> e;
    if (logp) { /* the exponents stayed in HBM: this item's, now */
        e.resize((size_t) cand);
        HIP_TRY(hipSetDevice(b->ctx->device));
        HIP_TRY(hipMemcpy(e.data(), b->pairLogp.p + b->hItems[(size_t) item].pairBase, (size_t) cand * sizeof(double),
                          hipMemcpyDeviceToHost));
    }
    long long kept = 0;
    for (long long k = 0; k < cand; k++) {
        const int v = b->hPost[o + k];
        if (v < 0) continue;
        triples[kept * 3] = v;
        if (b->compactPairs) {
            const unsigned xy = ((const unsigned *) b->hPacked)[o + k];
            triples[kept * 3 + 1] = xy & 0xFFFFu;
            triples[kept * 3 + 2] = xy >> 16;
        } else {
            triples[kept * 3 + 1] = b->hPacked[o + k].x;
            triples[kept * 3 + 2] = b->hPacked[o + k].y;
        }
        if (logp) logp[kept] = e[(size_t) k];
        kept++;
    }
    return CPECAN_OK;
}

int cpecan_hip_batch_fetch_totals(cpecan_batch *b, int64_t item, int64_t *xay, double *total,
                                  int64_t cap) {
    if (!b || item < 0 || item >= b->nItems) return fail(CPECAN_EINVAL, "bad argument");
    int rc = ensure_counts(b);
    if (rc) return rc;
    const DevItem &d = b->hItems[(size_t) item];
    long long n = b->hNTot[(size_t) item];
    if (n > d.totCap) return fail(CPECAN_EOVERFLOW, "totals overflow (%lld > %lld)", n, d.totCap);
    if (n > cap) return fail(CPECAN_EOVERFLOW, "need room for %lld totals", n);
    if (n == 0) return CPECAN_OK;
    if (xay)
        HIP_TRY(hipMemcpy(xay, b->totXay.p + d.totBase, (size_t) n * sizeof(long long), hipMemcpyDeviceToHost));
    if (total)
        HIP_TRY(hipMemcpy(total, b->totVal.p + d.totBase, (size_t) n * sizeof(double), hipMemcpyDeviceToHost));
    return CPECAN_OK;
}

int cpecan_hip_batch_expectations_device_ptr(cpecan_batch *b, void **devPtr, int64_t *nDoubles) {
    if (!b || !devPtr) return fail(CPECAN_EINVAL, "bad argument");
    *devPtr = (void *) b->expect.p;
    if (nDoubles) *nDoubles = (int64_t) b->expect.n;
    return CPECAN_OK;
}

int cpecan_hip_batch_fetch_expectations(cpecan_batch *b, int32_t modelId, double *out) {
    if (!b || !out || modelId < 0 || modelId >= b->nModels) return fail(CPECAN_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    HIP_TRY(hipMemcpy(out, b->expect.p + (size_t) modelId * b->expectLen, (size_t) b->expectLen * sizeof(double),
                      hipMemcpyDeviceToHost));
    return CPECAN_OK;
}

int cpecan_hip_batch_debug_cells(cpecan_batch *b, int64_t item, double *forward, double *backward,
                                 int64_t nCells) {
    if (!b || item < 0 || item >= b->nItems) return fail(CPECAN_EINVAL, "bad argument");
    if (!b->P.debug || b->kernel != CPECAN_KERNEL_GENERAL)
        return fail(CPECAN_EINVAL, "batch was not created with CPECAN_FLAG_DEBUG_DUMP");
    const DevItem &d = b->hItems[(size_t) item];
    if (nCells < d.nCells) return fail(CPECAN_EOVERFLOW, "need room for %lld cells", d.nCells);
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipStreamSynchronize(b->ctx->stream));
    if (forward)
        HIP_TRY(hipMemcpy(forward, b->Fstore.p + d.cellBase * 3, (size_t) d.nCells * 3 * sizeof(double),
                          hipMemcpyDeviceToHost));
    if (backward)
        HIP_TRY(hipMemcpy(backward, b->dbgB.p + d.cellBase * 3, (size_t) d.nCells * 3 * sizeof(double),
                          hipMemcpyDeviceToHost));
    return CPECAN_OK;
}

} /* extern "C" */

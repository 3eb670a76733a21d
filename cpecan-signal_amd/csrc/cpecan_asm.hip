/*
 * cpecan_asm.hip -- host side of the hand-scheduled assembly sweeps (asm/gen_sweeps.py): the code object (assembled
 * from asm/cpecan_sweeps_gfx950.s, embedded below) is loaded once per device, its kernels are launched through the
 * module API with one argument block; the two small kernels here write what a forward wave starts from.
 */
#include "cpecan_asm.h"

#include <mutex>

namespace {

const unsigned char SWEEPS_HSACO[] = {
#include "cpecan_sweeps_hsaco.inc"
};

struct PerDevice {
    bool tried = false, ok = false;
    hipModule_t module = nullptr;
    hipFunction_t forward = nullptr, backward = nullptr;
    double *coef = nullptr;
};
std::mutex g_lock;
PerDevice g_dev[64];

/* the logAdd table (impl/pairwiseAligner.c:238-249: four cubics, float literals) by n = ceil(2 d): 16 rows of c3, c2, c1, c0 */
void fill_coef(double *c) {
    const float t[16] = { -0.009350833524763f, 0.130659527668286f, 0.498799810682272f, 0.693203116424741f,
                          -0.014532321752540f, 0.139942324101744f, 0.495635523139337f, 0.692140569840976f,
                          -0.004605031767994f, 0.063427417320019f, 0.695956496475118f, 0.514272634594009f,
                          -0.000458661602210f, 0.009695946122598f, 0.930734667215156f, 0.168037164329057f };
    for (int l = 0; l < 64; l++) {
        const int n = l >> 2, piece = n <= 2 ? 0 : n <= 5 ? 1 : n <= 9 ? 2 : 3;
        c[l] = (double) t[piece * 4 + (l & 3)];
    }
}

int launch(hipFunction_t f, hipStream_t stream, const AsmArgs *args) {
    AsmArgs a = *args;
    size_t bytes = sizeof a;
    void *extra[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END };
    return hipModuleLaunchKernel(f, (unsigned) a.nItems, 1, 1, 64, 1, 1, 0, stream, nullptr, extra) == hipSuccess ? 0 : -1;
}

} // namespace

extern "C" int cpecan_asm_load(int device) {
    if (device < 0 || device >= 64) return -1;
    std::lock_guard<std::mutex> g(g_lock);
    PerDevice &d = g_dev[device];
    if (d.tried) return d.ok ? 0 : -1;
    d.tried = true;
    if (hipSetDevice(device) != hipSuccess) return -1;
    if (hipModuleLoadData(&d.module, SWEEPS_HSACO) != hipSuccess) return -1;
    if (hipModuleGetFunction(&d.forward, d.module, "cpecan_k_asm_forward_l3") != hipSuccess) return -1;
    if (hipModuleGetFunction(&d.backward, d.module, "cpecan_k_asm_backward_l3") != hipSuccess) {
        d.backward = nullptr;
        (void) hipGetLastError(); /* (a code object without the backward kernel is not an error) */
    }
    double c[64];
    fill_coef(c);
    if (hipMalloc((void **) &d.coef, sizeof c) != hipSuccess) return -1;
    if (hipMemcpy(d.coef, c, sizeof c, hipMemcpyHostToDevice) != hipSuccess) return -1;
    d.ok = true;
    return 0;
}

extern "C" const double *cpecan_asm_coef(int device) { return cpecan_asm_load(device) == 0 ? g_dev[device].coef : nullptr; }

extern "C" int cpecan_asm_launch_forward(int device, hipStream_t stream, const AsmArgs *args) {
    if (cpecan_asm_load(device) != 0) return -1;
    return launch(g_dev[device].forward, stream, args);
}

extern "C" int cpecan_asm_launch_backward(int device, hipStream_t stream, const AsmArgs *args) {
    if (cpecan_asm_load(device) != 0 || !g_dev[device].backward) return -1;
    return launch(g_dev[device].backward, stream, args);
}

/* Context [2] of an alignment: the registers of a forward wave that has done diagonal 0 -- every slot parked on the
 * "not a k-mer" row (column 0 scores that sentinel, impl/pairwiseAligner.c:314-318), the start vector in cell (0, 0)
 * (stateMachine.c:1168-1177), the slots of the k-mers that enter and leave next -- laid out as the assembly loads it
 * (gen_sweeps.py: CTX_*). */
extern "C" __global__ void cpecan_k_asm_ctx_init(const DevItem *items, long long nItems, char *ctx, long long ctxBytes,
                                                 double *ring, long long ringDoubles, int ringD) {
    const long long idx = blockIdx.x;
    if (idx >= nItems) return;
    const int lane = threadIdx.x;
    const DevItem it = items[idx];
    char *c = ctx + (idx * 3 + 2) * ctxBytes;
    const double ninf = CP_NEG_INF;
    for (int j = 0; j < ASM_L; j++)
        for (int q = 0; q < ASM_NCONST / 2; q++) { /* pairs (2q, 2q + 1) of the row: K1 = 3, K2 = 7, YK1 = 11, YK2 = 15, gap-X sums 16, 17 */
            double2 v;
            v.x = (2 * q == 16) ? ninf : 0.0;
            v.y = (2 * q + 1 == 3 || 2 * q + 1 == 7 || 2 * q + 1 == 11 || 2 * q + 1 == 15 || 2 * q + 1 == 17) ? ninf : 0.0;
            *(double2 *) (c + (long long) (9 * j + q) * 1024 + lane * 16) = v;
        }
    for (int p = 0; p < 2; p++)
        for (int j = 0; j < ASM_L; j++) {
            const bool origin = p == 0 && j == 0 && lane == 0; /* diagonal 0 is even: cell (0, 0) in slot 0 */
            char *x = c + ASM_CTX_X + (long long) (p * ASM_L + j) * 1536;
            double2 xy;
            xy.x = xy.y = origin && it.raggedL ? 0.0 : ninf;
            *(double *) (x + lane * 8) = origin && !it.raggedL ? 0.0 : ninf;
            *(double2 *) (x + 512 + lane * 16) = xy;
        }
    if (lane < 16) {
        /* masks: lane 0 of layer 0; the next k-mer (column 1) enters slot 1, the next to leave is slot 0 */
        const int s[16] = { 1, 0, 0, 0, 0, 0, 1 / ASM_L, 1 % ASM_L, 0, 0, 0, 0, 0, 0, 0, 0 };
        ((int *) (c + ASM_CTX_S))[lane] = s[lane];
    }
    /* the row lanes without a cell read on the way back: -inf everywhere */
    double *r = ring + idx * ringDoubles + (long long) ringD * (ASM_ROW_BYTES / 8);
    for (int i = lane; i < ASM_ROW_BYTES / 8; i += 64) r[i] = ninf;
}

/* The mask table of one batch: one thread per diagonal. */
extern "C" __global__ void cpecan_k_asm_masks(const DevItem *items, long long nItems, const int *bandTab, unsigned *maskTab) {
    const long long idx = blockIdx.y;
    if (idx >= nItems) return;
    const DevItem it = items[idx];
    const long long d = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (d > it.lX + it.lY) return;
    const int lo = bandTab[2 * (it.diagBase + d)], hi = bandTab[2 * (it.diagBase + d) + 1];
    unsigned long long m[ASM_L], g[ASM_L];
    for (int j = 0; j < ASM_L; j++) m[j] = g[j] = 0ull;
    for (int x = lo - 1; x <= hi + 1; x++) { /* (column -1: the last slot, parked while the band starts at column 0) */
        const int s = (x + 64 * ASM_L) % (64 * ASM_L);
        g[s % ASM_L] |= ((1ull << ASM_MASK_GROUP) - 1) << ((s / ASM_L) & ~(ASM_MASK_GROUP - 1)); /* whole 128-byte lines (every
                                                                 access of a ring row is 16 bytes per lane): no partial writes */
        if (x >= lo && x <= hi) m[s % ASM_L] |= 1ull << (s / ASM_L);
    }
    unsigned *e = maskTab + (it.diagBase + d) * (ASM_MASK_BYTES / 4);
    for (int j = 0; j < ASM_L; j++) {
        e[2 * j] = (unsigned) m[j];
        e[2 * j + 1] = (unsigned) (m[j] >> 32);
        e[8 + 2 * j] = (unsigned) g[j];
        e[9 + 2 * j] = (unsigned) (g[j] >> 32);
    }
    e[6] = (unsigned) lo;
    e[7] = (unsigned) hi;
    /* the last layer's 8-byte emissions: 16 lanes to a line */
    unsigned long long h = 0ull;
    for (int q = 0; q < 64; q += 16)
        if (g[ASM_L - 1] & (0xFFFFull << q)) h |= 0xFFFFull << q;
    e[14] = (unsigned) h;
    e[15] = (unsigned) (h >> 32);
}

extern "C" int cpecan_asm_launch_masks(hipStream_t stream, const DevItem *items, long long nItems, long long maxDiags,
                                       const int *bandTab, unsigned *maskTab) {
    hipLaunchKernelGGL(cpecan_k_asm_masks, dim3((unsigned) ((maxDiags + 255) / 256), (unsigned) nItems), dim3(256), 0, stream, items,
                       nItems, bandTab, maskTab);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

/* Ring row 0: cell (0, 0) as the forward kernel of cpecan_kernel_wave.hip leaves it. */
extern "C" __global__ void cpecan_k_asm_begin(const DevItem *items, long long nItems, double *ring, long long ringDoubles) {
    const long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nItems) return;
    const DevItem it = items[idx];
    double *r = ring + idx * ringDoubles;
    r[0] = it.raggedL ? CP_NEG_INF : 0.0;                                          /* Fm */
    r[1] = 0.0;                                                                    /* pm */
    r[ASM_OFF_PY / 8] = 0.0;                                                       /* py */
    r[ASM_OFF_FXY / 8] = r[ASM_OFF_FXY / 8 + 1] = it.raggedL ? 0.0 : CP_NEG_INF; /* Fx, Fy */
}

extern "C" int cpecan_asm_launch_ctx_init(hipStream_t stream, const DevItem *items, long long nItems, char *ctx,
                                          long long ctxBytes, double *ring, long long ringDoubles, int ringD) {
    hipLaunchKernelGGL(cpecan_k_asm_ctx_init, dim3((unsigned) nItems), dim3(64), 0, stream, items, nItems, ctx, ctxBytes,
                       ring, ringDoubles, ringD);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int cpecan_asm_launch_begin(hipStream_t stream, const DevItem *items, long long nItems, double *ring,
                                       long long ringDoubles) {
    hipLaunchKernelGGL(cpecan_k_asm_begin, dim3((unsigned) ((nItems + 255) / 256)), dim3(256), 0, stream, items, nItems,
                       ring, ringDoubles);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

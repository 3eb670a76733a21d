// What a vector memory instruction costs the wave that issues it, beside arithmetic (timing study, not part of the product):
// every wave loops { K fp64 FMAs; one VMEM instruction }, at 1, 4 and 8 waves per CU (1 workgroup per CU), for
// x4 / x2 stores, a store under an empty EXEC, x4 loads (never waited for inside the loop), to a 1 KB region of the
// wave's own that it writes again and again (L2 hits) or streaming through 7680-byte rows (HBM).
// hipcc --offload-arch=gfx950 -O2 tools/ubench_vmem.hip -o /tmp/ubench_vmem && /tmp/ubench_vmem
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 2000
#define FMA16 ".rept 16\n v_fma_f64 %0, %0, %1, %2\n .endr\n"

#define KERNEL(name, K_FMA, VMEM)                                                                      \
    __global__ void name(unsigned long long *out, char *buf, long long stride, int iters) {           \
        double a = threadIdx.x * 1e-9 + 1.0, b = 1.0000001, c = 1e-7;                                  \
        const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                          \
        char *p = buf + (long long) wave * (8 << 20) + (threadIdx.x & 63) * 16;                        \
        unsigned long long t0 = __builtin_readcyclecounter();                                          \
        for (int i = 0; i < iters; i++) {                                                              \
            asm volatile(K_FMA : "+v"(a) : "v"(b), "v"(c));                                            \
            VMEM;                                                                                      \
            p += stride;                                                                               \
            if (((i + 1) & 1023) == 0) p -= 1024 * stride;                                             \
        }                                                                                              \
        asm volatile("s_waitcnt vmcnt(0)");                                                            \
        unsigned long long t1 = __builtin_readcyclecounter();                                          \
        if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;                                              \
        if (a == 12345.678) buf[0] = 1;                                                                \
    }

#define ST4 asm volatile("global_store_dwordx4 %0, v[104:107], off" ::"v"(p) : "memory")
#define ST2 asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(a) : "memory")
#define ST0 asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0\n global_store_dwordx4 %0, v[104:107], off\n s_mov_b64 exec, s[20:21]" ::"v"(p) : "memory", "s20", "s21")
#define ST16 asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffff\n global_store_dwordx4 %0, v[104:107], off\n s_mov_b64 exec, s[20:21]" ::"v"(p) : "memory", "s20", "s21")
#define LD4 asm volatile("global_load_dwordx4 v[100:103], %0, off" ::"v"(p) : "memory", "v100", "v101", "v102", "v103")
#define NONE
// three 16-byte stores / loads that together cover 3 KB: each instruction a contiguous KB (lane * 16 in its own KB), or the
// three interleaved per lane (lane * 48 + 16 j: every instruction touches all 24 lines, a third of each)
#define ST4x3 asm volatile("global_store_dwordx4 %0, v[104:107], off\n global_store_dwordx4 %0, v[104:107], off offset:1024\n global_store_dwordx4 %0, v[104:107], off offset:2048" ::"v"(p) : "memory")
#define ST4x3S asm volatile("global_store_dwordx4 %0, v[104:107], off\n global_store_dwordx4 %0, v[104:107], off offset:16\n global_store_dwordx4 %0, v[104:107], off offset:32" ::"v"(p + (threadIdx.x & 63) * 32) : "memory")
#define LD4x3 asm volatile("global_load_dwordx4 v[100:103], %0, off\n global_load_dwordx4 v[108:111], %0, off offset:1024\n global_load_dwordx4 v[112:115], %0, off offset:2048" ::"v"(p) : "memory", "v100", "v101", "v102", "v103", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115")
#define LD4x3S asm volatile("global_load_dwordx4 v[100:103], %0, off\n global_load_dwordx4 v[108:111], %0, off offset:16\n global_load_dwordx4 v[112:115], %0, off offset:32" ::"v"(p + (threadIdx.x & 63) * 32) : "memory", "v100", "v101", "v102", "v103", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115")

KERNEL(k_none16, FMA16, NONE)
KERNEL(k_st4_16, FMA16, ST4)
KERNEL(k_st2_16, FMA16, ST2)
KERNEL(k_st0_16, FMA16, ST0)
KERNEL(k_st16_16, FMA16, ST16)
KERNEL(k_ld4_16, FMA16, LD4)
KERNEL(k_none64, FMA16 FMA16 FMA16 FMA16, NONE)
KERNEL(k_st4_64, FMA16 FMA16 FMA16 FMA16, ST4)
KERNEL(k_st0_64, FMA16 FMA16 FMA16 FMA16, ST0)
KERNEL(k_ld4_64, FMA16 FMA16 FMA16 FMA16, LD4)
KERNEL(k_st3_64, FMA16 FMA16 FMA16 FMA16, ST4x3)
KERNEL(k_st3s_64, FMA16 FMA16 FMA16 FMA16, ST4x3S)
KERNEL(k_ld3_64, FMA16 FMA16 FMA16 FMA16, LD4x3)
KERNEL(k_ld3s_64, FMA16 FMA16 FMA16 FMA16, LD4x3S)

struct K { const char *name; void (*fn)(unsigned long long *, char *, long long, int); };

int main() {
    unsigned long long *out;
    char *buf;
    const int maxWaves = 256 * 8;
    hipMalloc(&out, maxWaves * 8);
    if (hipMalloc(&buf, (size_t) maxWaves * (8 << 20)) != hipSuccess) { printf("no memory\n"); return 1; }
    K ks[] = { {"16 fma", k_none16}, {"16 fma + store x4", k_st4_16}, {"16 fma + store x2", k_st2_16}, {"16 fma + store x4 exec=0", k_st0_16},
               {"16 fma + store x4 16 lanes", k_st16_16}, {"16 fma + load x4", k_ld4_16}, {"64 fma", k_none64},
               {"64 fma + store x4", k_st4_64}, {"64 fma + store x4 exec=0", k_st0_64}, {"64 fma + load x4", k_ld4_64},
               {"64 fma + 3 stores, a KB each", k_st3_64}, {"64 fma + 3 stores, 48-byte lanes", k_st3s_64},
               {"64 fma + 3 loads, a KB each", k_ld3_64}, {"64 fma + 3 loads, 48-byte lanes", k_ld3s_64} };
    for (long long stride : {0ll, 7680ll}) {
        printf("stride %lld: cycles per iteration per wave, avg (max), at 1 / 4 / 8 waves per CU\n", stride);
        for (auto &k : ks) {
            printf("  %-28s", k.name);
            for (int w : {1, 4, 8}) {
                hipLaunchKernelGGL(k.fn, dim3(256), dim3(64 * w), 0, 0, out, buf, stride, 10);
                hipLaunchKernelGGL(k.fn, dim3(256), dim3(64 * w), 0, 0, out, buf, stride, ITERS);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(256 * w);
                hipMemcpy(h.data(), out, 256 * w * 8, hipMemcpyDeviceToHost);
                double s = 0, mx = 0;
                for (auto v : h) { s += (double) v; if ((double) v > mx) mx = (double) v; }
                printf("  %7.1f (%7.1f)", s / h.size() / ITERS, mx / ITERS);
            }
            printf("\n");
        }
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}

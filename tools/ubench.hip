// Issue-rate microbenchmarks for gfx950 (timing study for the systolic kernels; not part of the product).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 400

// Each kernel: one workgroup of `blockDim.x` threads per CU slot; reports s_memtime cycles for wave 0.
#define KERNEL(name, ...)                                                                          \
    __global__ void name(unsigned long long *out, double *sink, int iters) {                      \
        double a = threadIdx.x * 1e-9 + 1.0, b = 1.0000001, c = 1e-7, d0 = a + 1, d1 = a + 2,      \
               d2 = a + 3, d3 = a + 4;                                                             \
        int s0 = iters, s1 = 1, v0 = threadIdx.x, v1 = 3;                                         \
        __shared__ double lds[512];                                                                \
        lds[threadIdx.x & 511] = a;                                                                \
        __syncthreads();                                                                           \
        unsigned long long t0 = __builtin_readcyclecounter();                                      \
        for (int i = 0; i < iters; i++) { __VA_ARGS__ }                                                \
        unsigned long long t1 = __builtin_readcyclecounter();                                      \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                                           \
        if (a + d0 + d1 + d2 + d3 + s1 + v0 + v1 == 12345.678) sink[0] = a + s0;                   \
    }

// 64 dependent fma per iteration
KERNEL(k_fma_dep, asm volatile(".rept 64\n v_fma_f64 %0, %0, %1, %2\n .endr" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(k_add_dep, asm volatile(".rept 64\n v_add_f64 %0, %0, %1\n .endr" : "+v"(a) : "v"(c));)
KERNEL(k_mul_dep, asm volatile(".rept 64\n v_mul_f64 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));)
// 4 independent chains, 16 each = 64 ops
KERNEL(k_fma_ilp4, asm volatile(".rept 16\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n .endr"
                                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c));)
// f32 dependent for comparison
KERNEL(k_f32_dep, { float f = (float) a; asm volatile(".rept 64\n v_fma_f32 %0, %0, %0, %0\n .endr" : "+v"(f)); a = f; })
// max/min/cmp/cndmask mix as in logAdd (dependent)
KERNEL(k_max_dep, asm volatile(".rept 64\n v_max_f64 %0, %0, %1\n .endr" : "+v"(a) : "v"(b));)
KERNEL(k_cmp_cnd, asm volatile(".rept 32\n v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n .endr" : "+v"(a), "+v"(b), "+v"(v0) : "v"(v1) : "vcc");)
// SALU only, dependent
KERNEL(k_salu_dep, asm volatile(".rept 64\n s_add_u32 %0, %0, %1\n .endr" : "+s"(s0) : "s"(s1) : "scc");)
// VALU(f64 dependent) interleaved 1:1 with SALU
KERNEL(k_valu_salu, asm volatile(".rept 32\n v_add_f64 %0, %0, %2\n s_add_u32 %1, %1, 1\n .endr" : "+v"(a), "+s"(s0) : "v"(c) : "scc");)
// DPP wave_shr on 32-bit
KERNEL(k_dpp, asm volatile(".rept 64\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n .endr" : "+v"(v0));)
// LDS: broadcast 16-byte reads, dependent use
KERNEL(k_lds_b128, {
    double2 q;
    asm volatile(".rept 16\n ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n .endr" : "=v"(q) : "v"(v1 * 0));
    a += q.x;
})
// LDS reads, 4 in flight
KERNEL(k_lds_b128_x4, {
    double2 q0, q1, q2, q3;
    asm volatile(".rept 4\n ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)\n .endr"
                 : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(v1 * 0));
    a += q0.x + q1.x + q2.x + q3.x;
})
// taken branches
KERNEL(k_branch, asm volatile(".rept 16\n s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n s_nop 0\n s_nop 0\n1:\n .endr" :: "s"(s1) : "scc");)
// s_barrier
KERNEL(k_barrier, asm volatile(".rept 8\n s_barrier\n .endr" ::: "memory");)
// v_readlane + exec-masked move (install)
KERNEL(k_readlane, asm volatile(".rept 32\n v_readlane_b32 %1, %0, 3\n s_nop 3\n v_mov_b32 %0, %1\n .endr" : "+v"(v0), "+s"(s0));)

// rounding / conversion ops considered for the logAdd piece index
KERNEL(k_ceil_dep, asm volatile(".rept 64\n v_ceil_f64 %0, %0\n .endr" : "+v"(a));)
KERNEL(k_cvt_pair, asm volatile(".rept 32\n v_cvt_i32_f64 %1, %0\n v_cvt_f64_i32 %0, %1\n .endr" : "+v"(a), "+v"(v0));)
KERNEL(k_ldexp_dep, asm volatile(".rept 64\n v_ldexp_f64 %0, %0, 1\n .endr" : "+v"(a));)
KERNEL(k_min_i32, asm volatile(".rept 64\n v_min_i32 %0, %0, %1\n .endr" : "+v"(v0) : "v"(v1));)
// long straight-line bodies (several KB of code per iteration): instruction-fetch behaviour
KERNEL(k_long_valu, asm volatile(".rept 704\n v_fma_f64 %0, %0, %1, %2\n .endr" : "+v"(a) : "v"(b), "v"(c));)
// every 8th instruction a taken branch over 8 dead instructions (64 bytes: target in another fetch line)
KERNEL(k_long_branchy, asm volatile(".rept 88\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n s_cmp_eq_u32 %3, %3\n s_cbranch_scc1 1f\n .rept 8\n v_fma_f64 %0, %0, %1, %2\n .endr\n1:\n .endr" : "+v"(a) : "v"(b), "v"(c), "s"(s1) : "scc");)
// same, branch never taken
KERNEL(k_long_nottaken, asm volatile(".rept 88\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n s_cmp_lg_u32 %3, %3\n s_cbranch_scc1 1f\n 1:\n .endr" : "+v"(a) : "v"(b), "v"(c), "s"(s1) : "scc");)
// mixed VALU/SALU long body
KERNEL(k_long_mixed, asm volatile(".rept 176\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %0, %0, %2, %3\n s_add_u32 %1, %1, 1\n .endr" : "+v"(a), "+s"(s0) : "v"(b), "v"(c) : "scc");)

struct K { const char *name; void (*fn)(unsigned long long *, double *, int); int ops; };

int main() {
    unsigned long long *out;
    double *sink;
    hipMalloc(&out, 4096 * 8);
    hipMalloc(&sink, 64);
    float lastMs = 0; double lastCyc = 0;
    K ks[] = { {"fma_f64 dependent", k_fma_dep, 64}, {"add_f64 dependent", k_add_dep, 64},
               {"mul_f64 dependent", k_mul_dep, 64}, {"fma_f64 4 chains", k_fma_ilp4, 64},
               {"fma_f32 dependent", k_f32_dep, 64}, {"max_f64 dependent", k_max_dep, 64},
               {"cmp_f64+cndmask", k_cmp_cnd, 64}, {"s_add dependent", k_salu_dep, 64},
               {"add_f64+s_add 1:1", k_valu_salu, 64}, {"dpp wave_shr", k_dpp, 64},
               {"ds_read_b128 serial", k_lds_b128, 16}, {"ds_read_b128 x4", k_lds_b128_x4, 16},
               {"branch taken", k_branch, 16}, {"s_barrier", k_barrier, 8},
               {"readlane+mov", k_readlane, 32}, {"ceil_f64 dependent", k_ceil_dep, 64}, {"cvt i32<->f64", k_cvt_pair, 64},
               {"ldexp_f64 dependent", k_ldexp_dep, 64}, {"min_i32 dependent", k_min_i32, 64},
               {"long valu 704", k_long_valu, 704},
               {"long branchy 88x(6+br)", k_long_branchy, 704}, {"long not-taken", k_long_nottaken, 704},
               {"long mixed 3v+1s", k_long_mixed, 704} };
    // occupancies: waves per SIMD = (threads per WG / 64 / 4) * WGs per CU; we launch 256 WGs (1 per CU)
    // with 64*4*w threads, w = 1, 2, 4 waves per SIMD (1024 threads max => w <= 4)
    printf("%-22s %13s %13s %13s   (cycles per op per wave avg/max at 1/2/4 waves per SIMD)\n", "kernel", "w=1", "w=2", "w=4");
    for (auto &k : ks) {
        printf("%-22s", k.name);
        for (int w = 1; w <= 4; w *= 2) {
            int threads = 64 * 4 * w;
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(threads), 0, 0, out, sink, 10);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(threads), 0, 0, out, sink, ITERS);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            lastMs = ms;
            std::vector<unsigned long long> h(4096);
            hipMemcpy(h.data(), out, 4096 * 8, hipMemcpyDeviceToHost);
            double s = 0, mx = 0;
            int nw = threads / 64;
            for (int b = 0; b < 256; b++)
                for (int q = 0; q < nw; q++) { double v = (double) h[b * 16 + q]; s += v; if (v > mx) mx = v; }
            printf(" %6.2f/%6.2f", s / 256 / nw / ITERS / k.ops, mx / ITERS / k.ops);
            lastCyc = mx;
        }
        printf("   [w=4: %.0f counter ticks in %.3f ms => %.2f GHz]\n", lastCyc, lastMs, lastCyc / lastMs / 1e6);
    }
    return 0;
}

// Round-3 micro-questions for the assembly sweeps (timing study; not part of the product).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/ubench2.hip -o /tmp/ubench2 && /tmp/ubench2
//  1. does an fp64 VALU instruction whose EXEC has only the low 32 lanes set issue faster than a full one?
//  2. what does an LDS read beyond the workgroup's allocation return (the logAdd table index is not clamped when
//     the result is discarded anyway)?
//  3. cost of s_swappc_b64 call/return, of a not-taken branch, of s_bitcmp + s_cbranch
//  4. VALU + LDS mix of a logAdd at one and two waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 400

#define KERNEL(name, ...)                                                                          \
    __global__ void name(unsigned long long *out, double *sink, int iters) {                      \
        double a = threadIdx.x * 1e-9 + 1.0, b = 1.0000001, c = 1e-7;                              \
        int s0 = iters, s1 = 1, v0 = threadIdx.x, v1 = 3;                                         \
        __shared__ double lds[512];                                                                \
        lds[threadIdx.x & 511] = a;                                                                \
        __syncthreads();                                                                           \
        unsigned long long t0 = __builtin_readcyclecounter();                                      \
        for (int i = 0; i < iters; i++) { __VA_ARGS__ }                                            \
        unsigned long long t1 = __builtin_readcyclecounter();                                      \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;         \
        if (a + b + c + s1 + v0 + v1 == 12345.678) sink[0] = a + s0;                               \
    }

KERNEL(k_full, asm volatile(".rept 64\n v_fma_f64 %0, %0, %1, %2\n .endr" : "+v"(a) : "v"(b), "v"(c));)
KERNEL(k_half, asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffffffff\n .rept 64\n v_fma_f64 %0, %0, %1, %2\n .endr\n s_mov_b64 exec, s[20:21]"
                            : "+v"(a) : "v"(b), "v"(c) : "s20", "s21");)
KERNEL(k_quarter, asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffff\n .rept 64\n v_fma_f64 %0, %0, %1, %2\n .endr\n s_mov_b64 exec, s[20:21]"
                               : "+v"(a) : "v"(b), "v"(c) : "s20", "s21");)
KERNEL(k_hihalf, asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b32 exec_lo, 0\n .rept 64\n v_fma_f64 %0, %0, %1, %2\n .endr\n s_mov_b64 exec, s[20:21]"
                              : "+v"(a) : "v"(b), "v"(c) : "s20", "s21");)
// call / return
KERNEL(k_swappc, asm volatile("s_getpc_b64 s[20:21]\n s_add_u32 s20, s20, 1f-.\n s_addc_u32 s21, s21, 0\n s_branch 2f\n"
                              "1:\n v_add_f64 %0, %0, %1\n s_setpc_b64 s[22:23]\n 2:\n"
                              ".rept 16\n s_swappc_b64 s[22:23], s[20:21]\n .endr"
                              : "+v"(a) : "v"(c) : "s20", "s21", "s22", "s23", "scc");)
KERNEL(k_nottaken, asm volatile(".rept 16\n s_bitcmp1_b32 %0, 7\n s_cbranch_scc1 1f\n v_add_f64 %1, %1, %2\n 1:\n .endr" : : "s"(s1), "v"(a), "v"(c) : "scc");)
// a logAdd-like mix: 4 independent chains, each: max,min,add,add,ceil,cvt,lshl_add, 2 ds_read_b128, 3 mul, 4 add, cmp, 2 cndmask
#define LADD1(x, y, t0, t1, t2, q0, q1)                                                                   \
    "v_max_f64 " t0 ", " x ", " y "\n v_min_f64 " t1 ", " x ", " y "\n v_add_f64 " t2 ", " t0 ", -" t1 "\n"  \
    "v_add_f64 v[60:61], " t2 ", " t2 "\n v_ceil_f64 v[60:61], v[60:61]\n v_cvt_i32_f64 v60, v[60:61]\n"      \
    "v_lshl_add_u32 v60, v60, 5, %2\n ds_read_b128 " q0 ", v60\n ds_read_b128 " q1 ", v60 offset:16\n"
KERNEL(k_ladd_serial, {
    double r;
    asm volatile(".rept 8\n"
                 "v_max_f64 v[40:41], %0, %1\n v_min_f64 v[42:43], %0, %1\n v_add_f64 v[44:45], v[40:41], -v[42:43]\n"
                 "v_add_f64 v[46:47], v[44:45], v[44:45]\n v_ceil_f64 v[46:47], v[46:47]\n v_cvt_i32_f64 v46, v[46:47]\n"
                 "v_and_b32 v46, 15, v46\n v_lshlrev_b32 v46, 5, v46\n ds_read_b128 v[48:51], v46\n ds_read_b128 v[52:55], v46 offset:16\n"
                 "s_waitcnt lgkmcnt(1)\n v_mul_f64 v[56:57], v[48:49], v[44:45]\n v_add_f64 v[56:57], v[56:57], v[50:51]\n"
                 "v_mul_f64 v[56:57], v[56:57], v[44:45]\n s_waitcnt lgkmcnt(0)\n v_add_f64 v[56:57], v[56:57], v[52:53]\n"
                 "v_mul_f64 v[56:57], v[56:57], v[44:45]\n v_add_f64 v[56:57], v[56:57], v[54:55]\n v_add_f64 v[56:57], v[56:57], v[42:43]\n"
                 "s_mov_b32 s20, 0\n s_mov_b32 s21, 0x401e0000\n"
                 "v_cmp_gt_f64 vcc, s[20:21], v[44:45]\n v_cndmask_b32 v58, v40, v56, vcc\n v_cndmask_b32 v59, v41, v57, vcc\n"
                 "v_add_f64 %0, v[58:59], 0\n"
                 ".endr"
                 : "+v"(a) : "v"(b)
                 : "vcc", "s20", "s21", "v58", "v59", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
                   "v54", "v55", "v56", "v57");
    (void) r;
})

__global__ void k_lds_oob(double *out) {
    __shared__ double lds[128];
    lds[threadIdx.x & 127] = 1.0 + threadIdx.x;
    __syncthreads();
    double q0, q1, q2;
    unsigned a0 = 8 * 127, a1 = 100000u & ~31u, a2 = 0xFFFFFFE0u;
    asm volatile("ds_read_b64 %0, %3\n ds_read_b64 %1, %4\n ds_read_b64 %2, %5\n s_waitcnt lgkmcnt(0)"
                 : "=v"(q0), "=v"(q1), "=v"(q2) : "v"(a0), "v"(a1), "v"(a2));
    if (threadIdx.x == 0) { out[0] = q0; out[1] = q1; out[2] = q2; }
}

struct K { const char *name; void (*fn)(unsigned long long *, double *, int); int ops; };

int main() {
    unsigned long long *out;
    double *sink;
    hipMalloc(&out, 4096 * 8);
    hipMalloc(&sink, 64);
    K ks[] = { {"fma_f64 exec full", k_full, 64}, {"fma_f64 exec low 32", k_half, 64}, {"fma_f64 exec low 16", k_quarter, 64},
               {"fma_f64 exec high 32", k_hihalf, 64}, {"swappc call+ret+1valu", k_swappc, 16},
               {"bitcmp+cbranch n/t+valu", k_nottaken, 16}, {"logAdd serial", k_ladd_serial, 8} };
    printf("%-26s %13s %13s %13s   (cycles per op per wave avg/max at 1/2/4 waves per SIMD)\n", "kernel", "w=1", "w=2", "w=4");
    for (auto &k : ks) {
        printf("%-26s", k.name);
        for (int w = 1; w <= 4; w *= 2) {
            int threads = 64 * 4 * w;
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(threads), 0, 0, out, sink, 10);
            hipLaunchKernelGGL(k.fn, dim3(256), dim3(threads), 0, 0, out, sink, ITERS);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(4096);
            hipMemcpy(h.data(), out, 4096 * 8, hipMemcpyDeviceToHost);
            double s = 0, mx = 0;
            int nw = threads / 64;
            for (int b = 0; b < 256; b++)
                for (int q = 0; q < nw; q++) { double v = (double) h[b * 16 + q]; s += v; if (v > mx) mx = v; }
            printf(" %6.2f/%6.2f", s / 256 / nw / ITERS / k.ops, mx / ITERS / k.ops);
        }
        printf("\n");
    }
    double *o;
    hipMalloc(&o, 64);
    hipLaunchKernelGGL(k_lds_oob, dim3(1), dim3(64), 0, 0, o);
    double h[3];
    hipMemcpy(h, o, 24, hipMemcpyDeviceToHost);
    printf("lds in-range %.1f  beyond allocation %.17g  wrapped address %.17g  (%s)\n", h[0], h[1], h[2],
           hipGetErrorString(hipGetLastError()));
    return 0;
}

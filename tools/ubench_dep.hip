// Issue cost of fp64 vector instructions in a dependent chain against independent chains, one and two waves per SIMD
// (round 3: is a sweep bound by the number of vector instructions or by the length of its chains?).
// build and run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/ubench_dep.hip -o /tmp/ubd && /tmp/ubd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

template <int MODE> __global__ void k(double *out, long long *clk, int iters) {
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0000001, c = 0.5;
    float f0 = a0, f1 = 1.25f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) { REP32(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(b));) }
        if (MODE == 1) { REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (MODE == 2) { REP32(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));) }
        if (MODE == 3) { REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (MODE == 4) { REP32(asm volatile("v_ceil_f64 %0, %0" : "+v"(a0));) }
        if (MODE == 5) { REP8(asm volatile("v_ceil_f64 %0, %0\n v_ceil_f64 %1, %1\n v_ceil_f64 %2, %2\n v_ceil_f64 %3, %3"
                                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (MODE == 6) { REP32(asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));) }
        if (MODE == 7) { REP32(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(b));) }
        if (MODE == 8) { REP32(asm volatile("v_max_f64 %0, %0, %1" : "+v"(a0) : "v"(b));) }
        if (MODE == 9) { REP32(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f0) : "v"(f1));) }
        if (MODE == 10) { REP8(asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b));
                          asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(a0), "+v"(a1) : "v"(b));) }
        if (MODE == 11) { REP32(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f0) : "v"(a0)); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a0) : "v"(f0));) }
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + f0;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char *name, int perIter) {
    double *out; long long *clk;
    const int iters = 2000;
    for (int blocks : {1024, 2048}) {
        hipMalloc(&out, blocks * 64 * sizeof(double)); hipMalloc(&clk, blocks * sizeof(long long));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, clk, 10);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, clk, iters);
        hipDeviceSynchronize();
        std::vector<long long> h(blocks);
        hipMemcpy(h.data(), clk, blocks * sizeof(long long), hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        printf("%-34s %d waves per SIMD: %.2f cycles per instruction per wave\n", name, blocks / 1024, s / blocks / iters / perIter);
        hipFree(out); hipFree(clk);
    }
}

int main() {
    run<0>("v_add_f64 dependent chain", 32); run<10>("v_add_f64 two chains", 32); run<1>("v_add_f64 four chains", 32);
    run<2>("v_fma_f64 dependent chain", 32); run<3>("v_fma_f64 four chains", 32);
    run<7>("v_mul_f64 dependent chain", 32); run<8>("v_max_f64 dependent chain", 32);
    run<4>("v_ceil_f64 dependent chain", 32); run<5>("v_ceil_f64 four chains", 32);
    run<6>("v_add_f32 dependent chain", 32); run<9>("v_cndmask_b32 dependent chain", 32);
    run<11>("v_cvt_f32_f64 + v_cvt_f64_f32 chain", 64);
    return 0;
}

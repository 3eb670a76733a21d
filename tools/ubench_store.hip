// Store-path microbenchmark for gfx950: how many cycles a CU needs per coalesced global store
// instruction of 4/8/16 bytes per lane (timing study; not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITERS 1000

template <int W> __global__ void k_store(unsigned long long *out, char *buf, long long perWave, int iters, int stores) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long gw = (long long) blockIdx.x * (blockDim.x >> 6) + wave;
    char *p = buf + gw * perWave + lane * (4 * W);
    double v = lane;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        for (int s = 0; s < stores; s++) {
            char *q = p + ((long long) (i * stores + s) * (256 * W)) % perWave;
            if (W == 1) *(float *) q = (float) v;
            if (W == 2) *(double *) q = v;
            if (W == 4) *(double2 *) q = make_double2(v, v);
        }
        // some ALU work between bursts, as in the DP step (~200 dependent fma)
        asm volatile(".rept 200\n v_fma_f64 %0, %0, %0, %0\n .endr" : "+v"(v));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) out[gw] = t1 - t0;
}

template <int W> void run(const char *name, unsigned long long *out, char *buf, int wgs, int threads, int stores) {
    const long long perWave = 8ll << 20;
    hipLaunchKernelGGL(k_store<W>, dim3(wgs), dim3(threads), 0, 0, out, buf, perWave, 4, stores);
    hipLaunchKernelGGL(k_store<W>, dim3(wgs), dim3(threads), 0, 0, out, buf, perWave, ITERS, stores);
    hipDeviceSynchronize();
    int nw = wgs * threads / 64;
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), out, nw * 8, hipMemcpyDeviceToHost);
    double mx = 0, s = 0;
    for (auto v : h) { s += v; if (v > mx) mx = (double) v; }
    double bytes = (double) nw * ITERS * stores * 256.0 * W;
    printf("%-14s wgs %4d x %4d thr, %d stores/iter: cycles/iter avg %8.0f max %8.0f | %6.1f B/clk/CU | %.2f TB/s at 2.2GHz\n",
           name, wgs, threads, stores, s / nw / ITERS, mx / ITERS, bytes / mx / 256, bytes / mx * 2.2e9 / 1e12);
}

int main() {
    unsigned long long *out;
    char *buf;
    hipMalloc(&out, 65536 * 8);
    hipMalloc(&buf, (8ll << 20) * 4096 + 4096);
    for (int stores : {0, 5, 10}) {
        run<2>("dwordx2", out, buf, 256, 256, stores);   // 1 WG/CU
        run<2>("dwordx2", out, buf, 1024, 256, stores);  // 4 WG/CU
        run<4>("dwordx4", out, buf, 1024, 256, stores);
        run<1>("dword", out, buf, 1024, 256, stores);
    }
    return 0;
}

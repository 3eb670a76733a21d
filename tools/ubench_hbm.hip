// HBM bandwidth of the box at hand: streaming read, write and copy of 4 GiB (well past the 256 MiB Infinity Cache),
// best of 5, HIP-event timed.  Part of tools/box_diag.sh: boxes of the pool differ, and the wave-per-alignment sweeps
// (4 TB/s of HBM traffic while they run) are the ones that notice.
#include <hip/hip_runtime.h>
#include <cstdio>
#define BYTES (4ll << 30)
__global__ void k_read(const double2 *p, double *sink, long long n) {
    double acc = 0;
    for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345) sink[0] = acc;
}
__global__ void k_write(double2 *p, long long n) {
    for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) p[i] = make_double2((double) i, 1.0);
}
__global__ void k_copy(const double2 *p, double2 *q, long long n) {
    for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) q[i] = p[i];
}
template <typename F> static double best_ms(F launch) {
    hipEvent_t a, b;
    (void) hipEventCreate(&a);
    (void) hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        (void) hipEventRecord(a, 0);
        launch();
        (void) hipEventRecord(b, 0);
        (void) hipEventSynchronize(b);
        float ms = 0;
        (void) hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}
int main() {
    double2 *p, *q;
    double *sink;
    if (hipMalloc(&p, BYTES) != hipSuccess || hipMalloc(&q, BYTES) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void) hipMemset(p, 0, BYTES);
    (void) hipMemset(q, 0, BYTES);
    const long long n = BYTES / 16;
    const double rd = best_ms([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, p, sink, n); });
    const double wr = best_ms([&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, q, n); });
    const double cp = best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, p, q, n); });
    printf("hbm GB/s: read %.0f write %.0f copy(read+write) %.0f\n", BYTES / rd / 1e6, BYTES / wr / 1e6, 2.0 * BYTES / cp / 1e6);
    return 0;
}

// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for 8-byte-per-lane coalesced streaming accesses
// (the access width of the systolic kernels) against known byte counts.  Each kernel moves exactly
// BYTES bytes of a 4 GiB buffer (well past the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#define BYTES (4ll << 30)
__global__ void k_read8(const double *p, double *sink, long long n) {
    double acc = 0;
    for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) acc += p[i];
    if (acc == 1.2345) sink[0] = acc;
}
__global__ void k_read16(const double2 *p, double *sink, long long n) {
    double acc = 0;
    for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345) sink[0] = acc;
}
__global__ void k_write8(double *p, long long n) {
    for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) p[i] = (double) i;
}
__global__ void k_write16(double2 *p, long long n) {
    for (long long i = blockIdx.x * (long long) blockDim.x + threadIdx.x; i < n; i += (long long) gridDim.x * blockDim.x) p[i] = make_double2((double) i, 1.0);
}
int main() {
    double *buf, *sink;
    hipMalloc(&buf, BYTES);
    hipMalloc(&sink, 64);
    hipMemset(buf, 0, BYTES);
    hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, buf, sink, BYTES / 8);
    hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const double2 *) buf, sink, BYTES / 16);
    hipLaunchKernelGGL(k_write8, dim3(4096), dim3(256), 0, 0, buf, BYTES / 8);
    hipLaunchKernelGGL(k_write16, dim3(4096), dim3(256), 0, 0, (double2 *) buf, BYTES / 16);
    hipDeviceSynchronize();
    printf("each kernel moved %lld bytes\n", (long long) BYTES);
    return 0;
}

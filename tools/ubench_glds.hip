// What global_load_lds_dwordx4 does on gfx950 (the backward sweep's ring rows go to LDS with it): where the bytes land
// (M0, the instruction offset, the lane), which lanes write under a partial EXEC, and that vmcnt covers it.
// hipcc --offload-arch=gfx950 -O2 -o ubench_glds tools/ubench_glds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const unsigned *src, unsigned *out, int test) {
    __shared__ unsigned lds[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) lds[i] = 0xdead0000u + i;
    __syncthreads();
    const unsigned *p = src + lane * 4 + (test == 3 ? 1000 : 0);
    unsigned m0 = 1024;
    if (test == 0) {
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "s"(m0) : "memory");
    } else if (test == 1) {
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off offset:64\n\ts_waitcnt vmcnt(0)" ::"v"(p), "s"(m0) : "memory");
    } else if (test == 2) {
        if (lane & 1) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "s"(m0) : "memory");
    } else if (test == 3) {
        if (lane >= 16 && lane < 48) asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "s"(m0) : "memory");
    } else if (test == 4) {
        const unsigned *q = src + lane * 2;
        asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx3 %0, off\n\ts_waitcnt vmcnt(0)" ::"v"(q), "s"(m0) : "memory");
    }
    __syncthreads();
    for (int i = lane; i < 2048; i += 64) out[i] = lds[i];
}

int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = i;
    unsigned *src, *out;
    hipMalloc(&src, 4096 * 4);
    hipMalloc(&out, 2048 * 4);
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int test = 0; test < 5; test++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, test);
        std::vector<unsigned> o(2048);
        hipMemcpy(o.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
        printf("test %d:", test);
        int shown = 0;
        for (int i = 0; i < 2048 && shown < 40; i++)
            if (o[i] != 0xdead0000u + i) { printf(" [%d]=%u", i, o[i]); shown++; }
        int n = 0;
        for (int i = 0; i < 2048; i++) n += o[i] != 0xdead0000u + i;
        printf("  (%d dwords written)\n", n);
    }
    return 0;
}

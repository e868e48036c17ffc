// Diagnostic: what does the FIRST pass through a kernel's code cost? Kernels of N independent-register FMAs in
// straight-line code (8 bytes each), one pass (cold instruction fetch) or REP passes over the same code; one wave per
// SIMD. Build: hipcc -O2 --offload-arch=gfx950 -o tools/build/icache_probe tools/icache_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N>
__global__ void k_line(float* out, int reps) {
    float a = threadIdx.x, b = 1.0001f, c = 0.5f;
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    }
    if (a == 123.456f) out[0] = a;
}

template <int N>
static int run(float* d, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int reps : {1, 1, 1, 2, 4, 16}) {
        float best = 1e9f, first = 0;
        for (int it = 0; it < 6; ++it) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_line<N>, dim3(blocks), dim3(64), 0, 0, d, reps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it == 0) first = ms;
            if (ms < best) best = ms;
        }
        printf("code %6d B  blocks %5d  passes %2d: first %8.1f us  best %8.1f us   (execution alone: %.1f us)\n", N * 8, blocks, reps, first * 1e3, best * 1e3,
               double(N) * reps * 4 / 2.2e3);
    }
    return 0;
}

int main() {
    float* d; CK(hipMalloc(&d, 256));
    for (int blocks : {1, 1024, 6144}) {
        if (run<128>(d, blocks)) return 1;
        if (run<512>(d, blocks)) return 1;
        if (run<1024>(d, blocks)) return 1;
        if (run<4096>(d, blocks)) return 1;
    }
    return 0;
}

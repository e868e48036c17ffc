// Do the matrix pipe and the vector ALU of a gfx950 SIMD run side by side?
// Each wave loops over NM fp32 MFMAs (16x16x4, 8 passes = 32 cycles each) and NP packed fp32 FMAs
// (4 cycles each), interleaved, all independent of each other inside one iteration. If the pipes overlap,
// the time of the mixed loop is max(MFMA time, VALU time), not the sum.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/build/mfma_valu tools/mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int ITERS = 4096;

template <int NM, int NP>
__global__ void __launch_bounds__(64) k_mix(float* out, float a, float b) {
    v4f acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4f{a + i, b, a, b + threadIdx.x};
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = v2f{a * i, b + i};
    const v2f va = {a, a * 1.0001f}, vb = {b, b * 0.999f};
    const float fa = a + threadIdx.x, fb = b - threadIdx.x;
    for (int it = 0; it < ITERS; ++it) {
        constexpr int STEPS = NM > 0 ? NM : 1;
#pragma unroll
        for (int m = 0; m < STEPS; ++m) {
            if (NM > 0) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[m & 7], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NP / STEPS; ++q) {
                const int i = (m * (NP / STEPS) + q) & 15;
                asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(p[i]) : "s"(va), "v"(vb));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NM, int NP>
static void run(int cus, int waves_per_simd, float* out) {
    const int blocks = cus * 4 * waves_per_simd;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix<NM, NP>), dim3(blocks), dim3(64), 0, 0, out, 1.0f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<NM, NP>), dim3(blocks), dim3(64), 0, 0, out, 1.0f, 0.5f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: waves_per_simd waves x ITERS iterations
    const double ns_iter = ms * 1e6 / (double(ITERS) * waves_per_simd);
    const double mfma_tf = double(NM) * 2 * 16 * 16 * 4 * ITERS * blocks / (ms * 1e-3) / 1e12;
    const double pk_tf = double(NP) * 4 * 64 * ITERS * blocks / (ms * 1e-3) / 1e12;
    printf("waves/SIMD %d  MFMA %2d  pk_fma %3d : %8.3f ms  %7.1f ns per wave-iteration  MFMA %6.1f TFLOP/s  pk %6.1f TFLOP/s  sum %6.1f\n",
           waves_per_simd, NM, NP, ms, ns_iter, mfma_tf, pk_tf, mfma_tf + pk_tf);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s  CUs %d\n", p.gcnArchName, cus);
    float* out; CK(hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8));
    for (int w : {1, 2, 4, 6}) {
        switch (w) {
#define ALL(W) case W: run<8, 0>(cus, W, out); run<0, 64>(cus, W, out); run<8, 16>(cus, W, out); run<8, 32>(cus, W, out); \
                       run<8, 64>(cus, W, out); run<8, 96>(cus, W, out); run<4, 64>(cus, W, out); break;
            ALL(1) ALL(2) ALL(4) ALL(6)
        }
    }
    return 0;
}

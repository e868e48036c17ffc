// FP32 VALU issue-rate microbenchmark for MI355X (gfx950).
// Register-only chains of the instruction forms the sphere scan can be built from, at 1..8
// waves per SIMD. Reports, per form: wall TFLOP/s, shader cycles per wave-instruction per SIMD
// (from s_memtime, so independent of the clock the chip holds) and the effective clock.
// Answers SURVEY.md 7.3 item 2: which form reaches the 157.3 TFLOP/s "FP32 vector" figure.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/build/valu_peak tools/valu_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int ITERS = 1 << 14;
constexpr int ILP = 16;

#define KERNEL_HEAD(name)                                                                     \
    __global__ void __launch_bounds__(256) name(float* out, unsigned long long* cyc, float a, float b) { \
        float acc[ILP];                                                                       \
        _Pragma("unroll") for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x * 0.001f + i;    \
        float2v acc2[ILP];                                                                    \
        _Pragma("unroll") for (int i = 0; i < ILP; ++i) acc2[i] = float2v{acc[i], -acc[i]};   \
        float2v va = {a, a * 1.0001f}, vb = {b, b * 0.999f};                                  \
        (void)va; (void)vb;                                                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                 \
        for (int it = 0; it < ITERS; ++it) {                                                  \
            _Pragma("unroll") for (int i = 0; i < ILP; ++i) {

#define KERNEL_TAIL                                                                           \
            }                                                                                 \
        }                                                                                     \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                 \
        float s = 0.f;                                                                        \
        _Pragma("unroll") for (int i = 0; i < ILP; ++i) s += acc[i] + acc2[i].x + acc2[i].y;  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                       \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;      \
    }

KERNEL_HEAD(k_fma_vvv)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));   KERNEL_TAIL
KERNEL_HEAD(k_fma_svv)   asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(acc[i]) : "s"(a), "v"(b));   KERNEL_TAIL
KERNEL_HEAD(k_fmac_sv)   asm volatile("v_fmac_f32 %0, %1, %2"    : "+v"(acc[i]) : "s"(a), "v"(b));   KERNEL_TAIL
KERNEL_HEAD(k_fmac_vv)   asm volatile("v_fmac_f32 %0, %1, %2"    : "+v"(acc[i]) : "v"(a), "v"(b));   KERNEL_TAIL
KERNEL_HEAD(k_mul_sv)    asm volatile("v_mul_f32 %0, %1, %0"     : "+v"(acc[i]) : "s"(a));           KERNEL_TAIL
KERNEL_HEAD(k_sub_sv)    asm volatile("v_sub_f32 %0, %1, %0"     : "+v"(acc[i]) : "s"(a));           KERNEL_TAIL
KERNEL_HEAD(k_cmp_sv)    asm volatile("v_cmp_gt_f32 vcc, %1, %0" : "+v"(acc[i]) : "s"(a) : "vcc");   KERNEL_TAIL
KERNEL_HEAD(k_pk_vvv)    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(va), "v"(vb)); KERNEL_TAIL
KERNEL_HEAD(k_pk_svv)    asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(acc2[i]) : "s"(va), "v"(vb)); KERNEL_TAIL
// the scan's own shape, acc = s * ray + acc, and the same with the ray operand splat from ONE register (op_sel_hi: both halves
// take the low word): does reading fewer vector registers per instruction issue faster?
KERNEL_HEAD(k_pk_sva)    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[i]) : "s"(va), "v"(vb)); KERNEL_TAIL
KERNEL_HEAD(k_pk_sva_splat) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc2[i]) : "s"(va), "v"(vb)); KERNEL_TAIL
KERNEL_HEAD(k_pk_mul_sv) asm volatile("v_pk_mul_f32 %0, %1, %0"  : "+v"(acc2[i]) : "s"(va));         KERNEL_TAIL
KERNEL_HEAD(k_pk_add_sv) asm volatile("v_pk_add_f32 %0, %1, %0"  : "+v"(acc2[i]) : "s"(va));         KERNEL_TAIL
// other instructions of the sample loop (issue cost only; operands are whatever the registers hold)
KERNEL_HEAD(k_max_vv)    asm volatile("v_max_f32 %0, %0, %1"     : "+v"(acc[i]) : "v"(b));           KERNEL_TAIL
KERNEL_HEAD(k_max3_vvv)  asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));  KERNEL_TAIL
KERNEL_HEAD(k_cmp_vv)    asm volatile("v_cmp_ge_f32 vcc, %0, %1" : "+v"(acc[i]) : "v"(b) : "vcc");   KERNEL_TAIL
KERNEL_HEAD(k_cndmask)   asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(b) : "vcc"); KERNEL_TAIL
KERNEL_HEAD(k_cndmask_s)  { unsigned long long m = 0x5555555555555555ull ^ (unsigned long long)__builtin_amdgcn_readfirstlane(it);
                          asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(b), "s"(m)); } KERNEL_TAIL
KERNEL_HEAD(k_xor)       asm volatile("v_xor_b32 %0, %0, %1"     : "+v"(acc[i]) : "v"(b));           KERNEL_TAIL
KERNEL_HEAD(k_sqrt)      asm volatile("v_sqrt_f32 %0, %0"        : "+v"(acc[i]));                    KERNEL_TAIL
KERNEL_HEAD(k_rsq)       asm volatile("v_rsq_f32 %0, %0"         : "+v"(acc[i]));                    KERNEL_TAIL
KERNEL_HEAD(k_sin)       asm volatile("v_sin_f32 %0, %0"         : "+v"(acc[i]));                    KERNEL_TAIL
KERNEL_HEAD(k_mul_lo)    asm volatile("v_mul_lo_u32 %0, %0, %1"  : "+v"(acc[i]) : "v"(b));           KERNEL_TAIL
KERNEL_HEAD(k_mul_hi)    asm volatile("v_mul_hi_u32 %0, %0, %1"  : "+v"(acc[i]) : "v"(b));           KERNEL_TAIL
KERNEL_HEAD(k_mad_u64)   asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "+v"(acc2[i]) : "v"(a), "v"(b) : "vcc"); KERNEL_TAIL
KERNEL_HEAD(k_mul_u24)   asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(acc[i]) : "v"(b));           KERNEL_TAIL
KERNEL_HEAD(k_mulhi_u24) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(acc[i]) : "v"(b));        KERNEL_TAIL
KERNEL_HEAD(k_mad_u24)   asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b)); KERNEL_TAIL
KERNEL_HEAD(k_mov)       asm volatile("v_mov_b32 %0, %1"         : "+v"(acc[i]) : "v"(b));           KERNEL_TAIL
KERNEL_HEAD(k_sub_vv)    asm volatile("v_sub_f32 %0, %0, %1"     : "+v"(acc[i]) : "v"(b));           KERNEL_TAIL
KERNEL_HEAD(k_add_u32)   asm volatile("v_add_u32 %0, %0, %1"     : "+v"(acc[i]) : "v"(b));           KERNEL_TAIL
KERNEL_HEAD(k_min_u32)   asm volatile("v_min_u32 %0, %0, %1"     : "+v"(acc[i]) : "v"(b));           KERNEL_TAIL
KERNEL_HEAD(k_lshr)      asm volatile("v_lshrrev_b32 %0, 8, %0"  : "+v"(acc[i]));                    KERNEL_TAIL
KERNEL_HEAD(k_alignbit)  asm volatile("v_alignbit_b32 %0, %0, %1, 9" : "+v"(acc[i]) : "v"(b));       KERNEL_TAIL
KERNEL_HEAD(k_bitop3)    asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(acc[i]) : "v"(a), "v"(b)); KERNEL_TAIL
KERNEL_HEAD(k_cvt_u32)   asm volatile("v_cvt_f32_u32 %0, %0"     : "+v"(acc[i]));                    KERNEL_TAIL
KERNEL_HEAD(k_cmp_u32)   asm volatile("v_cmp_le_u32 vcc, %0, %1" : "+v"(acc[i]) : "v"(b) : "vcc");   KERNEL_TAIL
KERNEL_HEAD(k_fract)     asm volatile("v_fract_f32 %0, %0"       : "+v"(acc[i]));                    KERNEL_TAIL
KERNEL_HEAD(k_mul_f64)   asm volatile("v_mul_f64 %0, %0, %0"     : "+v"(acc2[i]));                   KERNEL_TAIL
KERNEL_HEAD(k_fma_f64)   asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(acc2[i]));                   KERNEL_TAIL
KERNEL_HEAD(k_add_f64)   asm volatile("v_add_f64 %0, %0, %0"     : "+v"(acc2[i]));                   KERNEL_TAIL
KERNEL_HEAD(k_cvt_f64)   asm volatile("v_cvt_f64_f32 %0, %1"     : "+v"(acc2[i]) : "v"(b));          KERNEL_TAIL
KERNEL_HEAD(k_rsq_f64)   asm volatile("v_rsq_f64 %0, %0"         : "+v"(acc2[i]));                   KERNEL_TAIL

typedef void (*kern_t)(float*, unsigned long long*, float, float);
struct Form { const char* name; kern_t k; int flops_per_lane; };

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    printf("device %s  CUs %d  max clock %.0f MHz\n", p.gcnArchName, cus, p.clockRate / 1000.0);
    float* out; CK(hipMalloc(&out, sizeof(float) * 256 * cus * 8));
    unsigned long long* cyc; CK(hipMalloc(&cyc, sizeof(unsigned long long) * cus * 8 * 4));
    std::vector<unsigned long long> hc(cus * 8 * 4);
    Form forms[] = {
        {"v_fma_f32 v,v,v", k_fma_vvv, 2}, {"v_fma_f32 s,v,v", k_fma_svv, 2}, {"v_fmac_f32 s,v", k_fmac_sv, 2},
        {"v_fmac_f32 v,v", k_fmac_vv, 2}, {"v_mul_f32 s,v", k_mul_sv, 1}, {"v_sub_f32 s,v", k_sub_sv, 1},
        {"v_cmp_gt_f32 s,v", k_cmp_sv, 1}, {"v_pk_fma_f32 v,v,v", k_pk_vvv, 4}, {"v_pk_fma_f32 s,v,v", k_pk_svv, 4}, {"v_pk_fma_f32 s,v,acc", k_pk_sva, 4}, {"v_pk_fma_f32 s,v.lo splat,acc", k_pk_sva_splat, 4},
        {"v_pk_mul_f32 s,v", k_pk_mul_sv, 2}, {"v_pk_add_f32 s,v", k_pk_add_sv, 2},
        {"v_max_f32 v,v", k_max_vv, 1}, {"v_max3_f32 v,v,v", k_max3_vvv, 1}, {"v_cmp_ge_f32 v,v", k_cmp_vv, 1},
        {"v_cndmask_b32", k_cndmask, 1}, {"v_cndmask_b32 sgpr mask", k_cndmask_s, 1}, {"v_xor_b32", k_xor, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_rsq_f32", k_rsq, 1},
        {"v_sin_f32", k_sin, 1}, {"v_mul_lo_u32", k_mul_lo, 1}, {"v_mul_hi_u32", k_mul_hi, 1}, {"v_mad_u64_u32", k_mad_u64, 1},
        {"v_mul_u32_u24", k_mul_u24, 1}, {"v_mul_hi_u32_u24", k_mulhi_u24, 1}, {"v_mad_u32_u24", k_mad_u24, 1}, {"v_mov_b32", k_mov, 1},
        {"v_sub_f32 v,v", k_sub_vv, 1}, {"v_add_u32", k_add_u32, 1}, {"v_min_u32", k_min_u32, 1}, {"v_lshrrev_b32", k_lshr, 1},
        {"v_alignbit_b32", k_alignbit, 1}, {"v_bitop3_b32", k_bitop3, 1}, {"v_cvt_f32_u32", k_cvt_u32, 1}, {"v_cmp_le_u32", k_cmp_u32, 1},
        {"v_fract_f32", k_fract, 1}, {"v_mul_f64", k_mul_f64, 1},
        {"v_fma_f64", k_fma_f64, 2}, {"v_add_f64", k_add_f64, 1}, {"v_cvt_f64_f32", k_cvt_f64, 1}, {"v_rsq_f64", k_rsq_f64, 1},
    };
    printf("%-20s %-6s %-9s %-9s %-12s\n", "form", "w/SIMD", "ms", "TFLOP/s", "Ginst/s/SIMD");
    for (const Form& f : forms) {
        for (int k = 2; k <= 8; k *= 4) {           // k blocks of 256 threads per CU = k waves per SIMD
            int grid = cus * k;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            hipLaunchKernelGGL(f.k, dim3(grid), dim3(256), 0, 0, out, cyc, 0.999f, 0.001f);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(f.k, dim3(grid), dim3(256), 0, 0, out, cyc, 0.999f, 0.001f);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(hc.data(), cyc, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost));
            std::sort(hc.begin(), hc.begin() + grid * 4);
            double med_cyc = double(hc[grid * 2]);
            double inst = double(ITERS) * ILP;                  // per wave
            double cyc_per_inst_simd = med_cyc / (inst * k);    // k waves share a SIMD
            double flops = double(grid) * 256 * inst * f.flops_per_lane;
            (void)cyc_per_inst_simd;
            // wave-instructions per second per SIMD: the issue rate that matters for a VALU-bound kernel
            printf("%-20s %-6d %-9.3f %-9.2f %-12.3f\n", f.name, k, ms, flops / ms / 1e9,
                   double(grid) * 4 * inst / (ms * 1e-3) / (cus * 4.0) / 1e9);
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
    }
    CK(hipFree(out)); CK(hipFree(cyc));
    return 0;
}

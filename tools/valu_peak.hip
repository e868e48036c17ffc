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
// Do the vector register banks matter for the scan's FMA shape, acc = s * ray + acc? Sixteen accumulator pairs at
// v[0:1], v[4:5] ... (registers 0 and 1 modulo 4) and the ray operand either in a pair of the same residues (v[64:65]) or of
// the other two (v[66:67]). Hard-coded registers: the allocator must not choose.
#define BANK_KERNEL(name, MPAIR)                                                                                         \
    __global__ void __launch_bounds__(256) name(float* out, unsigned long long* cyc, float a, float b) {                 \
        float2v va = {a, a * 1.0001f};                                                                                   \
        float r;                                                                                                         \
        asm volatile("v_mov_b32 v64, %0\n\tv_mov_b32 v65, %0\n\tv_mov_b32 v66, %0\n\tv_mov_b32 v67, %0" :: "v"(b) : "v64", "v65", "v66", "v67"); \
        asm volatile("v_mov_b32 v0, %0\n\tv_mov_b32 v1, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %0\n\tv_mov_b32 v8, %0\n\tv_mov_b32 v9, %0\n\t" \
                     "v_mov_b32 v12, %0\n\tv_mov_b32 v13, %0\n\tv_mov_b32 v16, %0\n\tv_mov_b32 v17, %0\n\tv_mov_b32 v20, %0\n\tv_mov_b32 v21, %0\n\t" \
                     "v_mov_b32 v24, %0\n\tv_mov_b32 v25, %0\n\tv_mov_b32 v28, %0\n\tv_mov_b32 v29, %0\n\tv_mov_b32 v32, %0\n\tv_mov_b32 v33, %0\n\t" \
                     "v_mov_b32 v36, %0\n\tv_mov_b32 v37, %0\n\tv_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\t" \
                     "v_mov_b32 v48, %0\n\tv_mov_b32 v49, %0\n\tv_mov_b32 v52, %0\n\tv_mov_b32 v53, %0\n\tv_mov_b32 v56, %0\n\tv_mov_b32 v57, %0\n\t" \
                     "v_mov_b32 v60, %0\n\tv_mov_b32 v61, %0" :: "v"(a) : BANK_CLOBBERS);                                  \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                            \
        for (int it = 0; it < ITERS; ++it) {                                                                             \
            asm volatile("v_pk_fma_f32 v[0:1], %0, " MPAIR ", v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, " MPAIR ", v[4:5]\n\t"  \
                         "v_pk_fma_f32 v[8:9], %0, " MPAIR ", v[8:9]\n\tv_pk_fma_f32 v[12:13], %0, " MPAIR ", v[12:13]\n\t" \
                         "v_pk_fma_f32 v[16:17], %0, " MPAIR ", v[16:17]\n\tv_pk_fma_f32 v[20:21], %0, " MPAIR ", v[20:21]\n\t" \
                         "v_pk_fma_f32 v[24:25], %0, " MPAIR ", v[24:25]\n\tv_pk_fma_f32 v[28:29], %0, " MPAIR ", v[28:29]\n\t" \
                         "v_pk_fma_f32 v[32:33], %0, " MPAIR ", v[32:33]\n\tv_pk_fma_f32 v[36:37], %0, " MPAIR ", v[36:37]\n\t" \
                         "v_pk_fma_f32 v[40:41], %0, " MPAIR ", v[40:41]\n\tv_pk_fma_f32 v[44:45], %0, " MPAIR ", v[44:45]\n\t" \
                         "v_pk_fma_f32 v[48:49], %0, " MPAIR ", v[48:49]\n\tv_pk_fma_f32 v[52:53], %0, " MPAIR ", v[52:53]\n\t" \
                         "v_pk_fma_f32 v[56:57], %0, " MPAIR ", v[56:57]\n\tv_pk_fma_f32 v[60:61], %0, " MPAIR ", v[60:61]"     \
                         :: "s"(va) : BANK_CLOBBERS);                                                                    \
        }                                                                                                                \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                            \
        asm volatile("v_add_f32 %0, v0, v61" : "=v"(r) :: BANK_CLOBBERS);                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                  \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                 \
    }
#define BANK_CLOBBERS "v0", "v1", "v4", "v5", "v8", "v9", "v12", "v13", "v16", "v17", "v20", "v21", "v24", "v25", "v28", "v29", "v32", "v33", \
                      "v36", "v37", "v40", "v41", "v44", "v45", "v48", "v49", "v52", "v53", "v56", "v57", "v60", "v61", "v64", "v65", "v66", "v67"
BANK_KERNEL(k_bank_same, "v[64:65]")
BANK_KERNEL(k_bank_other, "v[66:67]")
// the same sixteen instructions on CONSECUTIVE accumulator pairs v[0:1], v[2:3] ... v[30:31] (what an allocator hands out)
#define SEQ_CLOBBERS "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", \
                     "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v64", "v65"
__global__ void __launch_bounds__(256) k_bank_seq(float* out, unsigned long long* cyc, float a, float b) {
    float2v va = {a, a * 1.0001f};
    float r;
    asm volatile("v_mov_b32 v64, %0\n\tv_mov_b32 v65, %0" :: "v"(b) : "v64", "v65");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[2:3], %0, v[64:65], v[2:3]\n\t"
                     "v_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\tv_pk_fma_f32 v[6:7], %0, v[64:65], v[6:7]\n\t"
                     "v_pk_fma_f32 v[8:9], %0, v[64:65], v[8:9]\n\tv_pk_fma_f32 v[10:11], %0, v[64:65], v[10:11]\n\t"
                     "v_pk_fma_f32 v[12:13], %0, v[64:65], v[12:13]\n\tv_pk_fma_f32 v[14:15], %0, v[64:65], v[14:15]\n\t"
                     "v_pk_fma_f32 v[16:17], %0, v[64:65], v[16:17]\n\tv_pk_fma_f32 v[18:19], %0, v[64:65], v[18:19]\n\t"
                     "v_pk_fma_f32 v[20:21], %0, v[64:65], v[20:21]\n\tv_pk_fma_f32 v[22:23], %0, v[64:65], v[22:23]\n\t"
                     "v_pk_fma_f32 v[24:25], %0, v[64:65], v[24:25]\n\tv_pk_fma_f32 v[26:27], %0, v[64:65], v[26:27]\n\t"
                     "v_pk_fma_f32 v[28:29], %0, v[64:65], v[28:29]\n\tv_pk_fma_f32 v[30:31], %0, v[64:65], v[30:31]"
                     :: "s"(va) : SEQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("v_add_f32 %0, v0, v31" : "=v"(r) :: SEQ_CLOBBERS);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// ... and with only FOUR accumulators (each instruction depends on the fourth before it, as in a chain-interleaved scan)
__global__ void __launch_bounds__(256) k_bank_dep4(float* out, unsigned long long* cyc, float a, float b) {
    float2v va = {a, a * 1.0001f};
    float r;
    asm volatile("v_mov_b32 v64, %0\n\tv_mov_b32 v65, %0" :: "v"(b) : "v64", "v65");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[8:9], %0, v[64:65], v[8:9]\n\tv_pk_fma_f32 v[12:13], %0, v[64:65], v[12:13]\n\t"
                     "v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[8:9], %0, v[64:65], v[8:9]\n\tv_pk_fma_f32 v[12:13], %0, v[64:65], v[12:13]\n\t"
                     "v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[8:9], %0, v[64:65], v[8:9]\n\tv_pk_fma_f32 v[12:13], %0, v[64:65], v[12:13]\n\t"
                     "v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[8:9], %0, v[64:65], v[8:9]\n\tv_pk_fma_f32 v[12:13], %0, v[64:65], v[12:13]"
                     :: "s"(va) : SEQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("v_add_f32 %0, v0, v13" : "=v"(r) :: SEQ_CLOBBERS);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
// ... and with TWO (each depends on the second before it: the scan's two chains)
__global__ void __launch_bounds__(256) k_bank_dep2(float* out, unsigned long long* cyc, float a, float b) {
    float2v va = {a, a * 1.0001f};
    float r;
    asm volatile("v_mov_b32 v64, %0\n\tv_mov_b32 v65, %0" :: "v"(b) : "v64", "v65");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        asm volatile("v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]\n\t"
                     "v_pk_fma_f32 v[0:1], %0, v[64:65], v[0:1]\n\tv_pk_fma_f32 v[4:5], %0, v[64:65], v[4:5]"
                     :: "s"(va) : SEQ_CLOBBERS);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("v_add_f32 %0, v0, v5" : "=v"(r) :: SEQ_CLOBBERS);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

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
        {"pk_fma s,v,acc same banks", k_bank_same, 4}, {"pk_fma s,v,acc other banks", k_bank_other, 4},
        {"pk_fma s,v,acc consecutive pairs", k_bank_seq, 4}, {"pk_fma s,v,acc 4 chains", k_bank_dep4, 4}, {"pk_fma s,v,acc 2 chains", k_bank_dep2, 4},
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

"""Writes tools/build/mfma_valu2.hip: loops whose bodies are ONE inline-asm block each (fixed registers, no
compiler scheduling in the way) mixing NM matrix instructions with NP packed fp32 FMAs, all independent.
Question: do the matrix pipe and the vector ALU of a gfx950 SIMD overlap? Usage: python3 tools/mfma_valu_gen.py && hipcc --offload-arch=gfx950 -O3 -o
tools/build/mfma_valu2 tools/build/mfma_valu2.hip"""
import os

KINDS = {
    "f32x4": ("v_mfma_f32_16x16x4_f32 a[{a}:{a3}], v2, v3, a[{a}:{a3}]", 4),
    "f16x16": ("v_mfma_f32_16x16x16_f16 a[{a}:{a3}], v[2:3], v[4:5], a[{a}:{a3}]", 16),
    "f16x32": ("v_mfma_f32_16x16x32_f16 a[{a}:{a3}], v[2:5], v[6:9], a[{a}:{a3}]", 32),
}
CASES = [(k, nm, np_) for k in KINDS for nm, np_ in ((8, 0), (0, 64), (8, 8), (8, 16), (8, 32), (8, 64), (4, 64))]

src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstdlib>',
       '#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)',
       'constexpr int ITERS = 4096;']
names = []
for kind, nm, np_ in CASES:
    ins, _ = KINDS[kind]
    body = []
    per = np_ // nm if nm else 0
    pk = 0
    for m in range(max(nm, 1)):
        if nm:
            body.append(ins.format(a=4 * m, a3=4 * m + 3))
        for q in range(per if nm else np_):
            r = 16 + 2 * (pk % 16)
            body.append("v_pk_fma_f32 v[%d:%d], v[10:11], v[%d:%d], v[12:13]" % (r, r + 1, r, r + 1))
            pk += 1
    clob = ['"v%d"' % i for i in range(2, 48)] + ['"a%d"' % i for i in range(32)]
    name = "k_%s_%d_%d" % (kind, nm, np_)
    names.append((name, kind, nm, np_))
    init = "\\n".join("v_mov_b32 v%d, 0" % i for i in range(2, 48)) + "\\n" + "\\n".join("v_accvgpr_write_b32 a%d, 0" % i for i in range(32))
    src.append('__global__ void __launch_bounds__(64) %s(float* out) {' % name)
    src.append('    asm volatile("%s" ::: %s);' % (init, ", ".join(clob)))
    src.append('    for (int it = 0; it < ITERS; ++it) asm volatile("%s" ::: %s);' % ("\\n".join(body), ", ".join(clob)))
    src.append('    float r; asm volatile("s_nop 7\\ns_nop 7\\nv_accvgpr_read_b32 %0, a0\\nv_add_f32 %0, %0, v16" : "=v"(r) :: "a0", "v16");')
    src.append('    out[blockIdx.x * 64 + threadIdx.x] = r;')
    src.append('}')
src.append('typedef void (*kern_t)(float*);')
src.append('struct Case { const char* kind; int nm, np, kdepth; kern_t k; };')
src.append('static const Case cases[] = {')
for name, kind, nm, np_ in names:
    src.append('    {"%s", %d, %d, %d, %s},' % (kind, nm, np_, KINDS[kind][1], name))
src.append('};')
src.append(r'''
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s  CUs %d\n", p.gcnArchName, cus);
    float* out; CK(hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w : {1, 2, 4}) for (const Case& c : cases) {
        const int blocks = cus * 4 * w;
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double ns = ms * 1e6 / (double(ITERS) * w);
        const double mtf = double(c.nm) * 2 * 16 * 16 * c.kdepth * ITERS * blocks / (ms * 1e-3) / 1e12;
        const double ptf = double(c.np) * 4 * 64 * ITERS * blocks / (ms * 1e-3) / 1e12;
        printf("waves/SIMD %d  %-7s MFMA %2d  pk_fma %3d : %7.1f ns per wave-iteration per SIMD   matrix %7.1f TFLOP/s  packed %6.1f TFLOP/s\n",
               w, c.kind, c.nm, c.np, ns, mtf, ptf);
    }
    return 0;
}''')
os.makedirs("tools/build", exist_ok=True)
open("tools/build/mfma_valu2.hip", "w").write("\n".join(src) + "\n")

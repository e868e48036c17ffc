// pt_device.h — device functions of the sample loop for gfx950 (MI355X, CDNA4).
//
// One lane owns one path. A ray is (origin o, UNIT direction dh, len) with the reference's
// un-normalised direction d = len * dh, because the reference measures t in units of |d|
// (Sphere.cpp:7-17, t_min = 0.001 at Path Tracer.cpp:92) and reads |d| in the dielectric
// cosine (Dielectric.cpp:19). PhysicsVector<3> becomes three fp32 registers.
//
// The closest-hit search (HittableList.cpp:21-36 over Sphere.cpp:5-23) is split in three:
//   tier 1  every sphere, packed fp32 FMA on sphere PAIRS held in scalar registers. The ray's LINE is
//           described by dh and a point p of it near the world origin (the foot point
//           p = o - (o.dh) dh, on the line to an ulp of |p|; its small residual e = p.dh is kept); then
//           T = r^2 - dist(c, line)^2 = ((c-p).dh)^2 + 2 c.p - |p|^2 - K,   K = |c|^2 - r^2,
//           in which the sphere data only multiplies per-ray constants: a = c.dh - e, U = a*a - K + 2 c.p
//           (seven packed FMAs per pair; the per-ray |p|^2 is left to the compare);
//           T >= 0  <=>  the line meets the sphere (disc >= 0 at Sphere.cpp:10-13). Using p instead of
//           o keeps the magnitudes at the scale of the scene however far away the ray starts (with o
//           the discriminant of a small sphere seen from 300 units away drowns in |o|^2 ~ 1e5);
//   tier 2  only for the spheres some lane flagged: roots a -+ sqrt(T) (distances along the line
//           measured from p; the ray's own t is that minus o.dh, the same shift for every sphere),
//           the reference's root choice and range test (Sphere.cpp:16-23), closest-so-far update
//           with ties to the later index (HittableList.cpp:28-31);
//   finish  once per ray, for the winning sphere only, in fp64 from the double-precision scene:
//           t, point, outward normal, front face (Sphere.cpp:26-30, Hittable.h:26-31).
// The sphere the ray starts on (`prev`) is left out of tier 2 and handled in closed form: its
// near root is the start point itself, its far root is 2 (c-o).dh.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

#ifdef PT_STATS
extern __device__ unsigned long long g_stats[16];
#endif

namespace ptd {

typedef float v2f __attribute__((ext_vector_type(2)));

struct f3 { float x, y, z; };
PT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }   // VectorFunc.cpp:22-28
PT_DEV float dot(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
PT_DEV f3 fma3(f3 a, float s, f3 c) {   // a*s + c
    return mk3(__builtin_fmaf(a.x, s, c.x), __builtin_fmaf(a.y, s, c.y), __builtin_fmaf(a.z, s, c.z));
}
PT_DEV f3 select3(bool c, f3 a, f3 b) { return mk3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }

// ---------------------------------------------------------------------------------------------
// Scene and camera as the kernels see them.
// ---------------------------------------------------------------------------------------------
struct GeomRec {          // double-precision sphere for the finish (Sphere.h:20-21)
    double cx, cy, cz, r, inv_r, r2;
};
struct MatRec {           // 32 bytes, two float4 loads
    float ar, ag, ab, fuzz;        // Lambertian.h:18, Metal.h:17-18; (1, 1, 1) for a Dielectric, whose attenuation that is (Dielectric.cpp:7)
    float ior, inv_ior, r0;        // Dielectric.h:21; r0 = ((1-ior)/(1+ior))^2, Dielectric.cpp:47-48
    int kind;
};
struct SceneDev {
    const float* scan;     // n_pad/2 pair records of 8 floats: cx0,cx1, cy0,cy1, cz0,cz1, -K0,-K1  (K = |c|^2 - r^2)
    const GeomRec* geom;   // n
    const MatRec* mat;     // n
    int n;                 // spheres in list order
    int n_pad;             // multiple of 4; the padding never hits; two more never-hit quads follow it
};
struct CameraDev {         // Camera.cpp:27-31, as floats; ll_rel = lowerLeft - position
    f3 pos, ll_rel, hor, ver, u, v;
    float lens;
};

// ---------------------------------------------------------------------------------------------
// Random numbers: philox4x32-10, one block per (pixel, sample, bounce). Replaces RNG-A
// (Path Tracer.cpp:48-52), RNG-B (VectorFunc.cpp:15-19) and RNG-C (Dielectric.cpp:52-56).
// ---------------------------------------------------------------------------------------------
struct u4 { uint32_t x, y, z, w; };
// a ^ b ^ c in one instruction (v_bitop3_b32 with the truth table of a three-way xor); the compiler leaves two.
PT_DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
PT_DEV u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = uint64_t(0xD2511F53u) * c0;
        uint64_t p1 = uint64_t(0xCD9E8D57u) * c2;
        uint32_t n0 = xor3(uint32_t(p1 >> 32), c1, k0);
        uint32_t n2 = xor3(uint32_t(p0 >> 32), c3, k1);
        c1 = uint32_t(p1);
        c3 = uint32_t(p0);
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    u4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}
PT_DEV float u24(uint32_t w) { return float(w >> 8) * (1.0f / 16777216.0f); }   // exact, in [0, 1)

// sin / cos of 2*pi*x on the hardware unit (v_sin_f32 / v_cos_f32 take revolutions).
PT_DEV float sin_rev(float x) { return __builtin_amdgcn_sinf(x); }
PT_DEV float cos_rev(float x) { return __builtin_amdgcn_cosf(x); }
PT_DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
PT_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// VectorFunc.cpp:37-43: theta = 2 pi u1, phi = acos(1 - 2 u2), (sin phi cos theta, sin phi sin theta,
// cos phi). sin(acos z) = sqrt(1 - z^2), so no acos is needed.
PT_DEV f3 sphere_point(float u1, float u2) {
    float z = __builtin_fmaf(-2.0f, u2, 1.0f);
    float s = fast_sqrt(__builtin_fmaxf(__builtin_fmaf(-z, z, 1.0f), 0.0f));
    return mk3(s * cos_rev(u1), s * sin_rev(u1), z);
}

// Path Tracer.cpp:270-271 and Camera.cpp:38-46, jitter and unit-disk point given.
PT_DEV void gen_ray(const CameraDev& cam, int i, int j, float inv_w1, float inv_h1, float u1, float u2, float diskx,
                    float disky, f3& o, f3& d) {
    float s = (float(i) + u1) * inv_w1;
    float t = (float(j) + u2) * inv_h1;
    float rx = diskx * cam.lens, ry = disky * cam.lens;            // Camera.cpp:40
    f3 off = fma3(cam.u, rx, cam.v * ry);                          // :42
    o = cam.pos + off;                                             // :45
    d = fma3(cam.hor, s, fma3(cam.ver, t, cam.ll_rel)) - off;      // :45
}

// Path Tracer.cpp:105-107 from the unit direction's y.
PT_DEV f3 sky(float dy) {
    float t = 0.5f * (dy + 1.0f);
    float w = 1.0f - t;
    return mk3(__builtin_fmaf(0.5f, t, w), __builtin_fmaf(0.7f, t, w), w + t);
}

// ---------------------------------------------------------------------------------------------
// Tier 2 for one flagged sphere: Sphere.cpp:16-23 and HittableList.cpp:28-31 in unit-direction t.
// ---------------------------------------------------------------------------------------------
// T = r^2 - dist(c, line)^2, b = position of the sphere's centre along the line (measured from p), tmin likewise.
// The closest hit so far is kept as best_d = its distance BEYOND tmin (>= 0, +inf for none). The two roots beyond tmin
// are b' -+ sqrt(T) with b' = b - tmin; the reference takes the near one unless it lies before tmin, then the far one,
// and rejects that too if it lies before tmin (Sphere.cpp:17-21): "the smallest non-negative of the two". Negative
// floats (and NaNs, for T < 0: Sphere.cpp:13) compare LARGER than every non-negative float when their bits are read as
// unsigned integers, so that choice is one unsigned minimum, and "within [0, best_d]" one unsigned compare.
PT_DEV void tier2(float T, float b, int j, int prev, float tmin, float& best_d, int& best_i) {
#ifdef PT_STATS
    if (threadIdx.x == 0) atomicAdd(&g_stats[8], 1ull);               // spheres whose roots were taken (wave level)
#endif
    asm volatile("" : "+s"(j));   // keep j a scalar register: hipcc otherwise rewrites `j != prev` into a vector add per quad in tier 1
    const float s = fast_sqrt(T);
    const float bp = b - tmin;
    const float t1 = bp - s, t2 = bp + s;                                            // Sphere.cpp:17, :19
    uint32_t tu = min(__float_as_uint(t1), __float_as_uint(t2));
    const bool ok = (j != prev) & (tu <= __float_as_uint(best_d));                   // :18-21; <= : later index wins ties
    // Two moves under the execution mask instead of two selects and a scalar-to-vector copy of j (the empty asm keeps
    // hipcc from turning the branch back into selects): one vector instruction fewer per flagged sphere.
    if (ok) {
        asm volatile("" : "+v"(tu));
        best_d = __uint_as_float(tu);
        best_i = j;
    }
}

typedef float v16f __attribute__((ext_vector_type(16)));

// One s_load_dwordx16. The destination counts as written when the statement ends although the data
// arrives later: swait() on the same value must come before its first use.
PT_DEV v16f sload16(const float* p) {
    v16f r;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(r) : "s"(p));
    return r;
}
PT_DEV void swait(v16f& r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r)); }

// Tier 1 and, when flagged, tier 2 for the four spheres of one record quad R (spheres j .. j+3).
// dh: unit direction; ne = -p.dh, p2 = 2 p, pp = |p|^2 with p the foot point of the line; tmin is a distance along
// the line measured from p, best_d the closest hit's distance beyond tmin (see tier2).
PT_DEV void scan_quad(const v16f& R, int j, f3 dh, float ne, f3 p2, float pp, int prev, float tmin, float& best_d,
                      int& best_i) {
    // Seven packed FMAs per sphere pair, each with exactly one scalar operand:
    //   a = c.dh - p.dh;  U = a*a + (-K) + 2 p.c  ( = T + |p|^2 );  the line meets the sphere iff U >= |p|^2.
    v2f B[2], U[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const v2f cx = v2f{R[8 * h + 0], R[8 * h + 1]}, cy = v2f{R[8 * h + 2], R[8 * h + 3]};
        const v2f cz = v2f{R[8 * h + 4], R[8 * h + 5]}, nK = v2f{R[8 * h + 6], R[8 * h + 7]};   // nK = r^2 - |c|^2
        v2f b = __builtin_elementwise_fma(cx, v2f{dh.x, dh.x}, v2f{ne, ne});
        b = __builtin_elementwise_fma(cy, v2f{dh.y, dh.y}, b);
        b = __builtin_elementwise_fma(cz, v2f{dh.z, dh.z}, b);
        v2f u = __builtin_elementwise_fma(b, b, nK);
        u = __builtin_elementwise_fma(cx, v2f{p2.x, p2.x}, u);
        u = __builtin_elementwise_fma(cy, v2f{p2.y, p2.y}, u);
        u = __builtin_elementwise_fma(cz, v2f{p2.z, p2.z}, u);
        B[h] = b;
        U[h] = u;
    }
    // (the first pair's maximum first: its chain ends one instruction earlier, so no wait state in front of the reduce)
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(U[0].x, U[0].y), U[1].x), U[1].y);
#ifdef PT_STATS
    if (threadIdx.x == 0) atomicAdd(&g_stats[2], 1ull);
    if (__any(m >= pp) && threadIdx.x == 0) atomicAdd(&g_stats[3], 1ull);
    { unsigned long long bm = __ballot(m >= pp); if (threadIdx.x == 0) atomicAdd(&g_stats[4], (unsigned long long)__popcll(bm)); }
#endif
    if (__any(m >= pp)) {                       // wave-uniform: some lane's line meets a sphere of this quad
        // one wave-uniform branch per sphere, so that the three spheres nobody flagged cost a compare each
        if (__any(U[0].x >= pp)) tier2(U[0].x - pp, B[0].x, j + 0, prev, tmin, best_d, best_i);
        if (__any(U[0].y >= pp)) tier2(U[0].y - pp, B[0].y, j + 1, prev, tmin, best_d, best_i);
        if (__any(U[1].x >= pp)) tier2(U[1].x - pp, B[1].x, j + 2, prev, tmin, best_d, best_i);
        if (__any(U[1].y >= pp)) tier2(U[1].y - pp, B[1].y, j + 3, prev, tmin, best_d, best_i);
    }
}

// The linear scan. o, dh: ray origin and unit direction; tmin in unit-direction units
// (0.001 * len); prev: sphere the ray starts on, or -1; self_t: the far root of that sphere
// (self_far_root below). Returns the approximate t and the index of the closest hit (-1: miss).
PT_DEV void scan_spheres(const SceneDev& sc, f3 o, f3 dh, float tmin, int prev, float self_t, float& best_t, int& best_i) {
    // The foot point of the line. o.dh carries a rounding error of about an ulp of |o|, which only moves p
    // ALONG the line; what it leaves behind, p.dh, is small, so it is known accurately and tier 1 subtracts it.
    const float od = dot(o, dh);
    const f3 p = fma3(dh, -od, o);
    const float ne = -dot(p, dh);
    const f3 p2 = p * 2.0f;
    const float pp = dot(p, p);
    const float shift = od;                      // distance from p to o along the line: t = tau - shift
    const float tmin_ray = tmin;
    tmin += shift;
    self_t += shift;
    float best_d = __builtin_inff();             // the closest hit's distance beyond tmin
    best_i = -1;
    // The scan records come through the scalar cache, one s_load_dwordx16 (four spheres) at a time, in two
    // register banks: the request for the next quad is issued right after the wait that delivers the
    // current one, so it is in flight for the whole of the current quad's arithmetic. (hipcc emits the
    // load next to its first use, which exposes the scalar-cache miss latency once the scene outgrows
    // that cache; the loads and their waits are therefore written out. SMEM returns out of order, so
    // the only usable wait is lgkmcnt(0), placed BEFORE the next request.)
    const float* rec = sc.scan;
    const int n_pad = sc.n_pad;                  // multiple of 4: the host pads, and two more never-hit quads follow
    v16f bankA = sload16(rec);
    for (int j = 0; j < n_pad; j += 8) {
        swait(bankA);
        v16f bankB = sload16(rec + 16);
        scan_quad(bankA, j, dh, ne, p2, pp, prev, tmin, best_d, best_i);
        swait(bankB);
        bankA = sload16(rec + 32);
        if (j + 4 < n_pad) scan_quad(bankB, j + 4, dh, ne, p2, pp, prev, tmin, best_d, best_i);
        rec += 32;
    }
    swait(bankA);                                // the last request (never-hit padding) must land before the registers are reused
    // The sphere the ray starts on: the near root is the start point (rejected by t_min in the
    // reference), the far root counts only when the ray heads into the sphere.
    {
        const float sd = self_t - tmin;
        bool ok = (prev >= 0) & (sd >= 0.0f) & (sd <= best_d);
        best_d = ok ? sd : best_d;
        best_i = ok ? prev : best_i;
    }
    best_t = best_d + tmin_ray;                  // back to the ray's own (approximate) t; inf stays inf
}

// The same search with the SPHERES spread over a GROUP of G lanes that all hold the same ray (every argument the
// same in the lanes of a group; a wave holds 64 / G rays): lane `sub` of the group tests quads sub, sub + G, ..., then
// the group's candidates are reduced to the closest hit, ties to the later index. The per-sphere arithmetic is that of
// scan_quad and tier2 operation for operation, and the rule "smallest t, then largest index" does not depend on the
// order of the comparisons, so the answer is bit for bit the one scan_spheres gives. Used by k_tail for the long
// paths left at the end of a frame, where latency counts and most of the chip is idle.
//
// Two phases per chunk of up to 32 quads per lane (round 3; the first version took the roots of every sphere): phase 1 is
// tier 1 only, seven packed FMAs per pair and one compare per quad, and notes the quads this lane's line meets in a bit
// mask; phase 2 revisits the noted quads (a line meets a handful of the spheres of a scene: most lanes note none, the
// loop runs once or twice per chunk for the wave) and takes the roots there. Two fifths of the instructions per bounce.
//
// The records come through an accessor: quad(q, R) fills R[16] with quad q's record (SceneDev::scan layout).
struct QuadsGlobal {                 // the scan records where pt_set_scene put them (global memory / L2)
    const float* base;
    PT_DEV void quad(int q, float (&R)[16]) const {
        const float4* rec = reinterpret_cast<const float4*>(base + 16 * size_t(q));
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
        R[0] = r0.x; R[1] = r0.y; R[2] = r0.z; R[3] = r0.w; R[4] = r1.x; R[5] = r1.y; R[6] = r1.z; R[7] = r1.w;
        R[8] = r2.x; R[9] = r2.y; R[10] = r2.z; R[11] = r2.w; R[12] = r3.x; R[13] = r3.y; R[14] = r3.z; R[15] = r3.w;
    }
};
struct QuadsLds {                    // k_tail's copy in LDS, TRANSPOSED: 16-byte word c (0..3) of quad q sits at [c * nq + q],
    const float4* base;              // so the G lanes of a group, which read consecutive quads, read consecutive 16-byte
    int nq;                          // words (no bank conflicts; with the record layout of global memory they were 64 bytes apart: 4-way)
    PT_DEV void quad(int q, float (&R)[16]) const {
        const float4 r0 = base[q], r1 = base[nq + q], r2 = base[2 * nq + q], r3 = base[3 * nq + q];
        R[0] = r0.x; R[1] = r0.y; R[2] = r0.z; R[3] = r0.w; R[4] = r1.x; R[5] = r1.y; R[6] = r1.z; R[7] = r1.w;
        R[8] = r2.x; R[9] = r2.y; R[10] = r2.z; R[11] = r2.w; R[12] = r3.x; R[13] = r3.y; R[14] = r3.z; R[15] = r3.w;
    }
};
// tier 1 of one quad as scan_quad computes it: b and U of its four spheres
PT_DEV void coop_tier1(const float (&R)[16], f3 dh, float ne, f3 p2, v2f (&B)[2], v2f (&U)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const v2f cx = v2f{R[8 * h + 0], R[8 * h + 1]}, cy = v2f{R[8 * h + 2], R[8 * h + 3]};
        const v2f cz = v2f{R[8 * h + 4], R[8 * h + 5]}, nK = v2f{R[8 * h + 6], R[8 * h + 7]};
        v2f b = __builtin_elementwise_fma(cx, v2f{dh.x, dh.x}, v2f{ne, ne});
        b = __builtin_elementwise_fma(cy, v2f{dh.y, dh.y}, b);
        b = __builtin_elementwise_fma(cz, v2f{dh.z, dh.z}, b);
        v2f u = __builtin_elementwise_fma(b, b, nK);
        u = __builtin_elementwise_fma(cx, v2f{p2.x, p2.x}, u);
        u = __builtin_elementwise_fma(cy, v2f{p2.y, p2.y}, u);
        u = __builtin_elementwise_fma(cz, v2f{p2.z, p2.z}, u);
        B[h] = b;
        U[h] = u;
    }
}
template <int G, class Quads>
PT_DEV void coop_scan(const Quads& recs, int n_pad, int sub, f3 o, f3 dh, float tmin, int prev, float self_t, float& best_t, int& best_i) {
    const float od = dot(o, dh);
    const f3 p = fma3(dh, -od, o);
    const float ne = -dot(p, dh);
    const f3 p2 = p * 2.0f;
    const float pp = dot(p, p);
    const float shift = od;
    const float tmin_ray = tmin;
    tmin += shift;
    self_t += shift;
    float best_d = __builtin_inff();
    best_i = -1;
    const int n_quads = n_pad >> 2;
    // the roots of quad q's four spheres from tier 1's b and U: the arithmetic of tier2, operation for operation
    auto roots = [&](int q, const v2f (&B)[2], const v2f (&U)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = 4 * q + 2 * h + k;
                const float s = fast_sqrt(U[h][k] - pp);                   // NaN for a line that misses: its bits are above +inf's
                const float bp = B[h][k] - tmin;
                const float t1 = bp - s, t2 = bp + s;
                const uint32_t tu = min(__float_as_uint(t1), __float_as_uint(t2));
                const bool ok = (j != prev) & (tu <= __float_as_uint(best_d));   // j rises within a lane: <= keeps the later index
                best_d = ok ? __uint_as_float(tu) : best_d;
                best_i = ok ? j : best_i;
            }
        }
    };
    if (G == 64 && n_quads <= 2 * G) {
        // A whole wave on one ray and a quad or two per lane: one pass, the roots straight from tier 1's registers where
        // the line meets a sphere of the quad (some lane's does in nearly every pass: two phases would only read and
        // compute the same quads twice).
        for (int q = sub; q < n_quads; q += G) {
            float R[16];
            recs.quad(q, R);
            v2f B[2], U[2];
            coop_tier1(R, dh, ne, p2, B, U);
            const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(U[0].x, U[0].y), U[1].x), U[1].y);
            if (m >= pp) roots(q, B, U);
        }
    } else {
        constexpr int CHUNK = 32;                                        // quads per lane and chunk: one bit each
        for (int q0 = sub; q0 < n_quads; q0 += G * CHUNK) {              // (the lanes of a wave differ in `sub` only: same trip count +- 1)
            // phase 1: which of my quads does the line meet?
            uint32_t noted = 0u;
            for (int i = 0; i < CHUNK && q0 + G * i < n_quads; ++i) {
                float R[16];
                recs.quad(q0 + G * i, R);
                v2f B[2], U[2];
                coop_tier1(R, dh, ne, p2, B, U);
                const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(U[0].x, U[0].y), U[1].x), U[1].y);
                noted |= (m >= pp ? 1u : 0u) << i;
            }
            // phase 2: the roots, for the noted quads only
            while (__any(noted != 0u)) {
                if (noted != 0u) {
                    const int i = __builtin_ctz(noted);
                    noted &= noted - 1u;
                    const int q = q0 + G * i;
                    float R[16];
                    recs.quad(q, R);
                    v2f B[2], U[2];
                    coop_tier1(R, dh, ne, p2, B, U);
                    roots(q, B, U);
                }
            }
        }
    }
    // The group's closest hit, in every lane of the group. "Smaller t, then larger index" is a total order: first the
    // group's smallest t (bit patterns of non-negative floats and +inf order like unsigned integers, as in the roots above),
    // then the largest index among the lanes that hold that t (-1: no hit, or not this lane's). A group of 16 lanes is a
    // DPP row: each of the four steps of either reduction is ONE instruction, a minimum or maximum whose second operand
    // comes through a row rotation (row_ror:8, 4, 2, 1) inside the vector ALU. (Round 3. The first version went through
    // the LDS crossbar, two round trips per step; the second compared (t, index) pairs, nine instructions per step; all
    // in the dependent chain of every bounce of the frame's longest path.)
    if constexpr (G == 4 || G == 16 || G == 64) {
        const uint32_t mine = __float_as_uint(best_d);
        uint32_t m = mine;
#define PT_ROR_MIN(CTRL) m = min(m, uint32_t(__builtin_amdgcn_update_dpp(int(0xffffffffu), int(m), CTRL, 0xf, 0xf, false)));
        if constexpr (G == 4) { PT_ROR_MIN(0xb1) PT_ROR_MIN(0x4e) }     // a group of four lanes: quad_perm:[1,0,3,2], then [2,3,0,1]
        else { PT_ROR_MIN(0x128) PT_ROR_MIN(0x124) PT_ROR_MIN(0x122) PT_ROR_MIN(0x121) }
#undef PT_ROR_MIN
        if constexpr (G == 64) {
            // the whole wave holds one ray (every lane enabled): each of the four rows has its minimum in all its lanes;
            // they meet through scalar registers (v_readlane of lanes 0, 16, 32, 48), no trip through LDS
            const uint32_t r0 = __builtin_amdgcn_readlane(int(m), 0), r1 = __builtin_amdgcn_readlane(int(m), 16);
            const uint32_t r2 = __builtin_amdgcn_readlane(int(m), 32), r3 = __builtin_amdgcn_readlane(int(m), 48);
            m = min(min(r0, r1), min(r2, r3));
        }
        int ci = mine == m ? best_i : -1;
#define PT_ROR_MAX(CTRL) ci = max(ci, __builtin_amdgcn_update_dpp(int(0x80000000u), ci, CTRL, 0xf, 0xf, false));
        if constexpr (G == 4) { PT_ROR_MAX(0xb1) PT_ROR_MAX(0x4e) }
        else { PT_ROR_MAX(0x128) PT_ROR_MAX(0x124) PT_ROR_MAX(0x122) PT_ROR_MAX(0x121) }
#undef PT_ROR_MAX
        if constexpr (G == 64) {
            const int r0 = __builtin_amdgcn_readlane(ci, 0), r1 = __builtin_amdgcn_readlane(ci, 16);
            const int r2 = __builtin_amdgcn_readlane(ci, 32), r3 = __builtin_amdgcn_readlane(ci, 48);
            ci = max(max(r0, r1), max(r2, r3));
        }
        best_d = __uint_as_float(m);
        best_i = ci;
    } else {
#pragma unroll
        for (int s = G / 2; s > 0; s >>= 1) {
            const float ot = __shfl_xor(best_d, s);
            const int oi = __shfl_xor(best_i, s);
            const bool take = (ot < best_d) | ((ot == best_d) & (oi > best_i));
            best_d = take ? ot : best_d;
            best_i = take ? oi : best_i;
        }
    }
    {
        const float sd = self_t - tmin;
        bool ok = (prev >= 0) & (sd >= 0.0f) & (sd <= best_d);
        best_d = ok ? sd : best_d;
        best_i = ok ? prev : best_i;
    }
    best_t = best_d + tmin_ray;
}

// Far root of the sphere a scattered ray starts on. The start point p lies on it, so
// c - p = -r * n_out with n_out = front ? n : -n, and the roots of the unit-direction quadratic
// are 0 and 2 (c - p).dh = -+ 2 r (n.dh).
PT_DEV float self_far_root(float r, f3 n, bool front, f3 dh_new) {
    const float nd = dot(n, dh_new);
    return (front ? -2.0f : 2.0f) * r * nd;
}

// ---------------------------------------------------------------------------------------------
// Finish: the hit record of the winning sphere in fp64 (Sphere.cpp:6-30, Hittable.h:26-31).
// t and tmin are in unit-direction units; the reference's t is t / len.
// ---------------------------------------------------------------------------------------------
// sqrt(x) for x >= 0 to ~1e-14 relative: the fp32 reciprocal square root as a seed, one Newton step in fp64
// (s = x r is off by d ~ 1e-7; s + r (x - s^2) / 2 is off by 1.5 d^2). A third of the instructions of the
// correctly rounded fp64 square root hipcc expands to; the finish needs 1e-9 at most.
PT_DEV double sqrt_refined(double x) {
    const float xf = float(x);
    const double r = double(__builtin_amdgcn_rsqf(xf));
    const double s = x * r;
    const double e = __builtin_fma(-s, s, x);
    const double s1 = __builtin_fma(0.5 * r, e, s);
    return xf > 1e-30f ? s1 : 0.0;               // a tangent hit: 0 (and no inf * 0)
}

struct HitRec {
    f3 p, n;
    bool front;
    float r;                         // radius of the sphere hit
    double t_unit;
    double px, py, pz, nx, ny, nz;   // the same point and normal before rounding to fp32
};
PT_DEV HitRec finish_hit(const GeomRec& g, bool self, f3 o, f3 dh, double tmin) {
    const double ox = o.x, oy = o.y, oz = o.z, dx = dh.x, dy = dh.y, dz = dh.z;
    const double ocx = ox - g.cx, ocy = oy - g.cy, ocz = oz - g.cz;            // Sphere.cpp:6
    const double a = dx * dx + dy * dy + dz * dz;                                // :7 (1 +- 2e-7)
    const double hb = ocx * dx + ocy * dy + ocz * dz;                            // :8
    const double cc = ocx * ocx + ocy * ocy + ocz * ocz - g.r2;                  // :9
    double disc = hb * hb - a * cc;                                              // :10
    disc = disc < 0.0 ? 0.0 : disc;      // tier 2 said "hit" in fp32; a grazing miss becomes a tangent hit
    const double sq = sqrt_refined(disc);                                        // :16
    const double inv_a = 2.0 - a;        // 1/a to 4e-14 since |a - 1| <= 2e-7
    const double t1 = (-hb - sq) * inv_a;                                        // :17
    const double t2 = (-hb + sq) * inv_a;                                        // :19
    const double t = (self || t1 < tmin) ? t2 : t1;                              // :18
    HitRec h;
    h.t_unit = t;
    h.r = float(g.r);
    h.px = ox + t * dx; h.py = oy + t * dy; h.pz = oz + t * dz;                  // :27, Ray.cpp:12-14
    double nx = (h.px - g.cx) * g.inv_r, ny = (h.py - g.cy) * g.inv_r, nz = (h.pz - g.cz) * g.inv_r;   // :28
    h.front = (dx * nx + dy * ny + dz * nz) < 0.0;                               // Hittable.h:28
    h.nx = h.front ? nx : -nx; h.ny = h.front ? ny : -ny; h.nz = h.front ? nz : -nz;   // :30
    h.p = mk3(float(h.px), float(h.py), float(h.pz));
    h.n = mk3(float(h.nx), float(h.ny), float(h.nz));
    return h;
}
PT_DEV HitRec finish_hit(const SceneDev& sc, int idx, bool self, f3 o, f3 dh, double tmin) {
    return finish_hit(sc.geom[idx], self, o, dh, tmin);
}

// ---------------------------------------------------------------------------------------------
// Scatter: Lambertian.cpp:5-16, Metal.cpp:6-15, Dielectric.cpp:5-50 computed side by side and
// blended with selects. dh, len: incoming ray; n: normal opposing the ray; u1,u2: sphere point;
// u3: Fresnel coin. out is the un-normalised scattered direction, like the reference's.
// ---------------------------------------------------------------------------------------------
PT_DEV bool shade(const MatRec& m, f3 dh, float len, f3 n, bool front, float u1, float u2, float u3, f3& out,
                  f3& atten) {
    const f3 S = sphere_point(u1, u2);                          // VectorFunc.cpp:37-43
    const float dn = dot(dh, n);                                // <= 0 by construction of n
    const f3 refl = fma3(n, -2.0f * dn, dh);                    // VectorFunc.cpp:66-68 on the unit vector

    // Lambertian.cpp:7-10
    // (the guard |lam| <= epsilon as a maximum norm: the sum of two unit vectors is that small only when their largest
    // components cancel exactly, and then every component is zero or tiny; one v_max3 with |.| modifiers instead of a dot product)
    f3 lam = n + S;
    lam = select3(__builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(lam.x), __builtin_fabsf(lam.y)), __builtin_fabsf(lam.z)) <= 1e-15f, n, lam);
    // Metal.cpp:10, :14. refl.n = dn (1 - 2 |n|^2) = -dn for the unit normal: "refl.n > 0" is "dn < 0", and exact grazing
    // (dn = 0) is refused either way; three instructions fewer than the dot product.
    const f3 met = fma3(S, m.fuzz, refl);
    const bool met_ok = dn < 0.0f;
    // Dielectric.cpp:9-37
    const float ratio = front ? m.inv_ior : m.ior;              // :9
    const float ct = __builtin_fminf(-dn * len, 1.0f);          // :19, the direction is NOT normalised there
    const float st = fast_sqrt(__builtin_fmaxf(__builtin_fmaf(-ct, ct, 1.0f), 0.0f));   // :20
    const bool forbidden = ratio * st > 1.0f;                   // :21
    const float w = 1.0f - ct;
    const float w2 = w * w;
    const float refl_p = __builtin_fmaf(1.0f - m.r0, w2 * w2 * w, m.r0);   // :46-50
    const bool do_reflect = forbidden | (refl_p > u3);          // :27, :30
    // VectorFunc.cpp:70-76
    const float ct2 = __builtin_fminf(-dn, 1.0f);
    const f3 perp = fma3(n, ct2, dh) * ratio;
    const float par = -fast_sqrt(__builtin_fabsf(1.0f - dot(perp, perp)));
    const f3 refr = fma3(n, par, perp);
    const f3 die = select3(do_reflect, refl, refr);

    const bool is_lam = m.kind == 0, is_met = m.kind == 1;
    out = select3(is_lam, lam, select3(is_met, met, die));
    atten = mk3(m.ar, m.ag, m.ab);   // Lambertian.cpp:14, Metal.cpp:11; a Dielectric's record holds (1, 1, 1) (Dielectric.cpp:7): see MatRec
    return is_met ? met_ok : true;
}

// Normalise a direction: returns the unit vector, writes the length.
PT_DEV f3 normalise(f3 d, float& len) {
    const float l2 = dot(d, d);
    const float inv = __builtin_amdgcn_rsqf(l2);
    len = l2 * inv;
    return d * inv;
}

}   // namespace ptd

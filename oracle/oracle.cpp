// oracle.cpp — CPU restatement of the reference's per-pixel sample loop, in double precision.
//
// TEST INFRASTRUCTURE ONLY. Nothing the product ships may import, link or call this file.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
// or as the reported CPU baseline, never as the thing measured or shipped.
//
// PARITY UNPINNED. The reference (/root/reference, DryPerspective/Path-Tracer) holds no tests,
// no golden vectors and no fixtures, and it does not compile as shipped: PhysicsVector.h and
// ConfigReader.h belong to the author's un-vendored "Basic-Utilities" library (README.md:16;
// no version is pinned anywhere). Writing stand-ins for those headers to build the reference
// is not allowed, so no output of the reference itself can pin this file. What pins it:
//   * the C++ standard's mt19937 known answer (10000th draw of a default engine = 4123659995),
//   * the Random123 known-answer vectors for philox4x32-10,
//   * the hand-derived known-answer vectors listed in SURVEY.md section 4 (tests/golden/kat.json),
//   * bit-exact agreement with libstdc++'s <random> for the uniform_real_distribution draws.
// Every function cites the reference file:line it follows.
//
// Build: see oracle/Makefile  (g++ -O2 -std=c++17 -shared -fPIC).
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------------------------
// 3-vector of double. Stands for dp::PhysicsVector<3> (missing header; API from the call sites
// listed in SURVEY.md 8(c)): elementwise + - , scaledBy, innerProduct, length, getUnitVector,
// vectorProduct. getUnitVector divides by the length (the original's rounding is unknowable).
// ---------------------------------------------------------------------------------------------
struct V3 {
    double x = 0, y = 0, z = 0;
};
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline V3 scaled(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline double len2(V3 a) { return dot(a, a); }
static inline double len(V3 a) { return std::sqrt(len2(a)); }
static inline V3 unit(V3 a) { double l = len(a); return {a.x / l, a.y / l, a.z / l}; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline V3 hadamard(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }   // VectorFunc.cpp:22-28

// ---------------------------------------------------------------------------------------------
// mt19937, written out (the reference uses std::mt19937 at Path Tracer.cpp:49, VectorFunc.cpp:16,
// Dielectric.cpp:54). Known answer: 10000th output of the default seed 5489 is 4123659995.
// ---------------------------------------------------------------------------------------------
struct MT19937 {
    uint32_t s[624];
    int idx = 624;
    explicit MT19937(uint32_t seed = 5489u) { reseed(seed); }
    void reseed(uint32_t seed) {
        s[0] = seed;
        for (int i = 1; i < 624; ++i) s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + uint32_t(i);
        idx = 624;
    }
    void refill() {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = s[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            s[i] = v;
        }
        idx = 0;
    }
    uint32_t next() {
        if (idx >= 624) refill();
        uint32_t y = s[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    // libstdc++ (g++ 11) std::generate_canonical<double, 53>: two 32-bit draws, low word first,
    // sum / 2^64 in double, and a result that rounds up to 1.0 is replaced by nextafter(1, 0).
    double canonical() {
        double lo = double(next());
        double hi = double(next());
        double r = (lo + hi * 4294967296.0) / 18446744073709551616.0;
        if (r >= 1.0) r = std::nextafter(1.0, 0.0);
        return r;
    }
    // std::uniform_real_distribution<double>{a, b}(engine): canonical * (b - a) + a.
    double uniform(double a, double b) { return canonical() * (b - a) + a; }
};

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011). Used only in "philox" uniform mode, to feed this double
// path with exactly the numbers the GPU kernels draw (north_star: "under injected identical
// random numbers").
// ---------------------------------------------------------------------------------------------
static inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = uint64_t(0xD2511F53u) * c0;
        uint64_t p1 = uint64_t(0xCD9E8D57u) * c2;
        uint32_t n0 = uint32_t(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = uint32_t(p1);
        uint32_t n2 = uint32_t(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = uint32_t(p0);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline double u24(uint32_t w) { return double(w >> 8) * (1.0 / 16777216.0); }   // 24-bit uniform in [0,1)

// ---------------------------------------------------------------------------------------------
// Scene: flat sphere list in insertion order (order matters for ties, HittableList.cpp:27-32).
// kind: 0 Lambertian, 1 Metal, 2 Dielectric.
// ---------------------------------------------------------------------------------------------
struct SphereRec {
    V3 c; double r;
    int kind; V3 albedo; double fuzz; double ior;
};
struct Scene { std::vector<SphereRec> s; };

// Path Tracer.cpp:182-260. thr_diffuse / thr_metal are 0.6 / 0.9 in the reference (:242, :247);
// half_extent is 10 (:218); count is 250 (:199).
static int generate_scene(Scene& sc, int count, uint32_t seedA, uint32_t seedB, double half_extent,
                          double thr_diffuse, double thr_metal) {
    MT19937 A(seedA), B(seedB);
    sc.s.clear();
    // :182-195, list order as added.
    sc.s.push_back({{2, 1, 4}, 1, 0, {0.9, 0.1, 0.1}, 0, 0});
    sc.s.push_back({{0, -1000, -1}, 1000, 0, {0.5, 0.5, 0.5}, 0, 0});
    sc.s.push_back({{0, 1, 2}, 1, 2, {1, 1, 1}, 0, 1.5});
    sc.s.push_back({{0, 1, -2}, 1, 1, {0.8, 0.8, 0.8}, 0.8, 0});
    sc.s.push_back({{2, 1, -6}, 1, 1, {0.8, 0.8, 0.8}, 0.0, 0});
    for (int i = 0; i < count; ++i) {
        double m = A.uniform(0, 1);             // :203
        double r = A.uniform(0.1, 0.4);         // :205
        V3 c;
        int tries = 0;
        bool placed = false;
        while (true) {                          // :213-231
            if (++tries > 50) break;            // :215-216
            double x = A.uniform(-half_extent, half_extent);   // :218, x before z
            double z = A.uniform(-half_extent, half_extent);
            c = {x, r, z};
            bool clipped = false;
            for (const SphereRec& o : sc.s) {   // :220-226
                if (len(c - o.c) < r + o.r) { clipped = true; break; }
            }
            if (clipped) continue;
            placed = true;
            break;
        }
        if (!placed) break;                     // :236 ends the whole generation
        SphereRec rec{c, r, 0, {0, 0, 0}, 0, 0};
        if (m < thr_diffuse) {                  // :242-245, three RNG-B draws (VectorFunc.cpp:34)
            double cr = B.uniform(0, 1), cg = B.uniform(0, 1), cb = B.uniform(0, 1);
            rec.kind = 0; rec.albedo = {cr, cg, cb};
        } else if (m < thr_metal) {             // :247-251
            double cr = B.uniform(0.6, 1), cg = B.uniform(0.6, 1), cb = B.uniform(0.6, 1);
            double fz = A.uniform(0, 1) / 2;    // :249
            rec.kind = 1; rec.albedo = {cr, cg, cb}; rec.fuzz = fz < 1 ? fz : 1;   // Metal.h:21
        } else {                                // :255
            rec.kind = 2; rec.albedo = {1, 1, 1}; rec.ior = 1.5;
        }
        sc.s.push_back(rec);                    // :259
    }
    return int(sc.s.size());
}

// ---------------------------------------------------------------------------------------------
// Camera. Camera.cpp:7-32 (setup) and :38-46 (ray). The seven derived values are laid out as
// cam[19] = position, lowerLeft, horizontal, vertical, u, v (3 each), lensRadius.
// ---------------------------------------------------------------------------------------------
struct Cam { V3 pos, ll, hor, ver, u, v; double lens; };

static Cam camera_setup(V3 pos, V3 look, V3 up, double aspect, double focal, double fov_deg, double aperture,
                        double focus) {
    const double pi = 3.14159265358979;          // Camera.h:45 (truncated on purpose)
    double theta = fov_deg * pi / 180.0;         // Camera.h:48
    double h = focal * std::tan(theta / 2);      // Camera.cpp:10
    double vh = 2 * h, vw = aspect * vh;         // :12-13
    Cam c;
    V3 w = unit(pos - look);                     // :18
    c.u = unit(cross(up, w));                    // :21
    c.v = unit(cross(w, c.u));                   // :24
    c.hor = scaled(scaled(c.u, vw), focus);      // :27
    c.ver = scaled(scaled(c.v, vh), focus);      // :28
    c.ll = pos - scaled(c.hor, 0.5) - scaled(c.ver, 0.5) - scaled(scaled(w, focal), focus);   // :29
    c.lens = aperture / 2;                       // :31
    c.pos = pos;
    return c;
}
static Cam cam_from(const double* a) {
    Cam c;
    c.pos = {a[0], a[1], a[2]}; c.ll = {a[3], a[4], a[5]}; c.hor = {a[6], a[7], a[8]};
    c.ver = {a[9], a[10], a[11]}; c.u = {a[12], a[13], a[14]}; c.v = {a[15], a[16], a[17]}; c.lens = a[18];
    return c;
}
static void cam_to(const Cam& c, double* a) {
    const V3* v[6] = {&c.pos, &c.ll, &c.hor, &c.ver, &c.u, &c.v};
    for (int i = 0; i < 6; ++i) { a[3 * i] = v[i]->x; a[3 * i + 1] = v[i]->y; a[3 * i + 2] = v[i]->z; }
    a[18] = c.lens;
}
struct Ray { V3 o, d; };
// Camera.cpp:40-45 with the unit-disk point (diskx, disky) injected.
static Ray camera_ray(const Cam& c, double s, double t, double diskx, double disky) {
    double rx = diskx * c.lens, ry = disky * c.lens;          // :40
    V3 off = scaled(c.u, rx) + scaled(c.v, ry);               // :42
    return {c.pos + off, c.ll + scaled(c.hor, s) + scaled(c.ver, t) - c.pos - off};   // :45
}

// ---------------------------------------------------------------------------------------------
// Intersection. Sphere.cpp:5-33, Hittable.h:26-31, HittableList.cpp:21-36.
// ---------------------------------------------------------------------------------------------
struct Hit { double t = 0; V3 p, n; bool front = false; int idx = -1; };

static bool sphere_hit(const SphereRec& s, const Ray& r, double t_min, double t_max, Hit& h) {
    V3 oc = r.o - s.c;                                  // :6
    double a = len2(r.d);                               // :7
    double hb = dot(oc, r.d);                           // :8
    double c = len2(oc) - s.r * s.r;                    // :9
    double disc = hb * hb - a * c;                      // :10
    if (disc < 0) return false;                         // :13
    double sq = std::sqrt(disc);                        // :16
    double t = (-hb - sq) / a;                          // :17
    if (t < t_min || t_max < t) {                       // :18
        t = (-hb + sq) / a;                             // :19
        if (t < t_min || t_max < t) return false;       // :20-21
    }
    h.t = t;                                            // :26
    h.p = r.o + scaled(r.d, t);                         // :27, Ray.cpp:12-14
    V3 on = scaled(h.p - s.c, 1 / s.r);                 // :28
    h.front = dot(r.d, on) < 0;                         // Hittable.h:28
    h.n = h.front ? on : -on;                           // Hittable.h:30
    return true;
}
static bool list_hit(const Scene& sc, const Ray& r, double t_min, double t_max, Hit& out) {
    Hit tmp;
    bool any = false;
    double closest = t_max;                             // HittableList.cpp:24
    for (size_t i = 0; i < sc.s.size(); ++i) {          // :27
        if (sphere_hit(sc.s[i], r, t_min, closest, tmp)) {   // :28
            any = true;
            closest = tmp.t;                            // :30
            out = tmp;                                  // :31
            out.idx = int(i);
        }
    }
    return any;
}

// ---------------------------------------------------------------------------------------------
// Scattering. VectorFunc.cpp:37-43 (sphere point), :66-76 (reflect, refract), Lambertian.cpp:5-16,
// Metal.cpp:6-15, Dielectric.cpp:5-50. The uniforms are arguments, so a caller can inject them.
// ---------------------------------------------------------------------------------------------
static V3 sphere_point(double u1, double u2) {          // VectorFunc.cpp:40-42, theta drawn first
    double theta = 2 * 3.14159265358979323846 * u1;
    double phi = std::acos(1 - 2 * u2);
    return {std::sin(phi) * std::cos(theta), std::sin(phi) * std::sin(theta), std::cos(phi)};
}
static V3 reflect(V3 v, V3 n) { return v - scaled(n, 2 * dot(v, n)); }   // VectorFunc.cpp:66-68
static V3 refract(V3 r, V3 n, double ratio) {                            // VectorFunc.cpp:70-76
    V3 ur = unit(r);
    double ct = std::fmin(-dot(ur, n), 1.0);
    V3 perp = scaled(ur + scaled(n, ct), ratio);
    V3 par = scaled(n, -std::sqrt(std::fabs(1.0 - len2(perp))));
    return par + perp;
}
static double schlick(double ct, double ior) {                           // Dielectric.cpp:46-50
    double r0 = (1 - ior) / (1 + ior);
    r0 *= r0;
    return r0 + (1 - r0) * std::pow(1 - ct, 5);
}
// Returns the isScattered bool. u1,u2 feed the sphere point (Lambertian, Metal), u3 the Fresnel coin.
static bool scatter(const SphereRec& m, const Ray& in, const Hit& h, double u1, double u2, double u3, V3& atten,
                    Ray& out) {
    if (m.kind == 0) {                                                   // Lambertian.cpp:5-16
        V3 d = h.n + unit(sphere_point(u1, u2));                         // :7, VectorFunc.cpp:53-55
        if (len(d) <= std::numeric_limits<double>::epsilon()) d = h.n;   // :10
        out = {h.p, d};
        atten = m.albedo;
        return true;
    }
    if (m.kind == 1) {                                                   // Metal.cpp:6-15
        V3 refl = reflect(unit(in.d), h.n);                              // :8
        out = {h.p, refl + scaled(sphere_point(u1, u2), m.fuzz)};        // :10
        atten = m.albedo;
        return dot(refl, h.n) > 0;                                       // :14, un-fuzzed vector
    }
    atten = {1, 1, 1};                                                   // Dielectric.cpp:7
    double ratio = h.front ? 1 / m.ior : m.ior;                          // :9
    double ct = std::fmin(dot(h.n, -in.d), 1.0);                         // :19, direction NOT normalised
    double st = std::sqrt(1 - ct * ct);                                  // :20
    bool forbidden = ratio * st > 1;                                     // :21
    bool fresnel = schlick(ct, m.ior) > u3;                              // :27
    V3 d = (forbidden || fresnel) ? reflect(unit(in.d), h.n) : refract(in.d, h.n, ratio);   // :30-37
    out = {h.p, d};
    return true;
}

// Where the uniforms of one path come from.
struct Sampler {
    // mode 0: the reference's three Mersenne Twisters (faithful draw order and counts).
    // mode 1: one philox4x32-10 block per (pixel, sample, bounce), the GPU's numbering.
    // mode 2: injected: the scatter at the k-th hit of the path takes (u1, u2, u3) = inject[3 k .. 3 k + 2] (orc_trace).
    int mode = 0;
    const double* inject = nullptr;
    MT19937* a = nullptr;            // RNG-A, Path Tracer.cpp:48-52 (thread_local in the reference)
    MT19937* b = nullptr;            // RNG-B, VectorFunc.cpp:15-19
    MT19937* c = nullptr;            // RNG-C, Dielectric.cpp:52-56
    std::mutex* lock = nullptr;      // VectorFunc.cpp:18 global mutex when RNG-B is shared by workers
    uint32_t key[2] = {0, 0};
    uint32_t pixel = 0, sample = 0;
    uint32_t blk[4];
    void block(uint32_t bounce) {
        uint32_t ctr[4] = {pixel, sample, bounce, 0};
        philox4x32_10(ctr, key, blk);
    }
};

// Path Tracer.cpp:82-108, recursive as in the reference. depth_index counts rays from 0.
static V3 radiance(const Scene& sc, const Ray& r, int depth, int ray_index, Sampler& sm, uint64_t& rays) {
    if (depth <= 0) return {0, 0, 0};                                    // :86
    ++rays;
    Hit h;
    if (list_hit(sc, r, 0.001, std::numeric_limits<double>::infinity(), h)) {   // :92
        const SphereRec& m = sc.s[h.idx];
        double u1 = 0, u2 = 0, u3 = 0;
        if (sm.mode == 0) {
            if (sm.lock) sm.lock->lock();                                // VectorFunc.cpp:39
            if (m.kind == 2) {
                u3 = sm.c->uniform(0.0, 1.0);                            // Dielectric.cpp:52-56, one draw per hit
            } else {                                                     // (unguarded race in the reference)
                u1 = sm.b->uniform(0.0, 1.0);                            // VectorFunc.cpp:40
                u2 = sm.b->uniform(0.0, 1.0);                            // :41
            }
            if (sm.lock) sm.lock->unlock();
        } else if (sm.mode == 2) {
            u1 = sm.inject[3 * ray_index]; u2 = sm.inject[3 * ray_index + 1]; u3 = sm.inject[3 * ray_index + 2];
        } else {
            sm.block(uint32_t(ray_index) + 1);
            u1 = u24(sm.blk[0]); u2 = u24(sm.blk[1]); u3 = u24(sm.blk[2]);
        }
        V3 atten; Ray out;
        if (scatter(m, r, h, u1, u2, u3, atten, out))                    // :96
            return hadamard(radiance(sc, out, depth - 1, ray_index + 1, sm, rays), atten);   // :97
        return {0, 0, 0};                                                // :100
    }
    V3 ud = unit(r.d);                                                   // :105
    double t = 0.5 * (ud.y + 1);                                         // :106
    return scaled(V3{1, 1, 1}, 1.0 - t) + scaled(V3{0.5, 0.7, 1.0}, t);  // :107
}

// One sample of pixel (i, j): Path Tracer.cpp:270-275 and Camera.cpp:38-46.
static V3 sample_pixel(const Scene& sc, const Cam& cam, int W, int H, int i, int j, int depth, Sampler& sm,
                       uint64_t& rays) {
    double jx, jy, dx, dy;
    if (sm.mode == 0) {
        jx = sm.a->uniform(0, 1);                                        // :270 drawn first
        jy = sm.a->uniform(0, 1);                                        // :271
        while (true) {                                                   // VectorFunc.cpp:57-64
            if (sm.lock) sm.lock->lock();                                // :33 (inside randVector)
            dx = sm.b->uniform(-1, 1);
            dy = sm.b->uniform(-1, 1);
            (void)sm.b->uniform(-1, 1);                                  // third draw, discarded by :60
            if (sm.lock) sm.lock->unlock();
            if (dx * dx + dy * dy >= 1) continue;                        // :61
            break;
        }
    } else {
        // GPU numbering: block (pixel, sample, 0): jitter x, jitter y, disk radius^2, disk angle.
        // The disk point is a polar map of two uniforms, not the reference's rejection loop: same
        // distribution, different mapping from uniforms to the point (documented deviation).
        sm.block(0);
        jx = u24(sm.blk[0]); jy = u24(sm.blk[1]);
        double rr = std::sqrt(u24(sm.blk[2])), ang = 2 * 3.14159265358979323846 * u24(sm.blk[3]);
        dx = rr * std::cos(ang); dy = rr * std::sin(ang);
    }
    double s = (double(i) + jx) / (double(W) - 1);                       // :270
    double t = (double(j) + jy) / (double(H) - 1);                       // :271
    Ray r = camera_ray(cam, s, t, dx, dy);                               // :273
    return radiance(sc, r, depth, 0, sm, rays);                          // :275
}

// Path Tracer.cpp:62-74 for one channel.
static int quantise(double sum, int spp) {
    double v = std::sqrt((1.0 / double(spp)) * sum);
    v = v < 0.0 ? 0.0 : (v > 0.999 ? 0.999 : v);
    return int(256 * v);
}

}   // namespace orc

// =============================================================================================
// C entry points (ctypes).
// =============================================================================================
using namespace orc;

extern "C" {

uint32_t orc_mt19937_nth(uint32_t seed, int n) {
    MT19937 g(seed);
    uint32_t v = 0;
    for (int i = 0; i < n; ++i) v = g.next();
    return v;
}
void orc_uniform_draws(uint32_t seed, double a, double b, int n, double* out) {
    MT19937 g(seed);
    for (int i = 0; i < n; ++i) out[i] = g.uniform(a, b);
}
void orc_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }

void* orc_scene_create(int n, const double* cx, const double* cy, const double* cz, const double* r, const int* kind,
                       const double* ar, const double* ag, const double* ab, const double* fuzz, const double* ior) {
    Scene* sc = new Scene;
    for (int i = 0; i < n; ++i)
        sc->s.push_back({{cx[i], cy[i], cz[i]}, r[i], kind[i], {ar[i], ag[i], ab[i]}, fuzz[i], ior[i]});
    return sc;
}
void orc_scene_free(void* s) { delete static_cast<Scene*>(s); }

// Returns the number of spheres placed (5 fixed + random); arrays must hold count + 5 entries.
int orc_generate_scene(int count, uint32_t seedA, uint32_t seedB, double half_extent, double thr_diffuse,
                       double thr_metal, double* cx, double* cy, double* cz, double* r, int* kind, double* ar,
                       double* ag, double* ab, double* fuzz, double* ior) {
    Scene sc;
    int n = generate_scene(sc, count, seedA, seedB, half_extent, thr_diffuse, thr_metal);
    for (int i = 0; i < n; ++i) {
        const SphereRec& s = sc.s[i];
        cx[i] = s.c.x; cy[i] = s.c.y; cz[i] = s.c.z; r[i] = s.r; kind[i] = s.kind;
        ar[i] = s.albedo.x; ag[i] = s.albedo.y; ab[i] = s.albedo.z; fuzz[i] = s.fuzz; ior[i] = s.ior;
    }
    return n;
}

void orc_camera_setup(const double* pos, const double* look, const double* up, double aspect, double focal,
                      double fov_deg, double aperture, double focus, double* cam19) {
    Cam c = camera_setup({pos[0], pos[1], pos[2]}, {look[0], look[1], look[2]}, {up[0], up[1], up[2]}, aspect, focal,
                         fov_deg, aperture, focus);
    cam_to(c, cam19);
}

// T5: rays for pixel coordinates (i, j), jitter (u1, u2) and an injected unit-disk point.
void orc_gen_ray(const double* cam19, int n, int W, int H, const int* pi, const int* pj, const double* u1,
                 const double* u2, const double* diskx, const double* disky, double* o, double* d) {
    Cam c = cam_from(cam19);
    for (int k = 0; k < n; ++k) {
        double s = (double(pi[k]) + u1[k]) / (double(W) - 1);
        double t = (double(pj[k]) + u2[k]) / (double(H) - 1);
        Ray r = camera_ray(c, s, t, diskx[k], disky[k]);
        o[3 * k] = r.o.x; o[3 * k + 1] = r.o.y; o[3 * k + 2] = r.o.z;
        d[3 * k] = r.d.x; d[3 * k + 1] = r.d.y; d[3 * k + 2] = r.d.z;
    }
}

// T3: closest hit of each ray. o, d, p, nrm are [n][3]. idx = -1 on a miss.
void orc_scan(void* scene, int n, const double* o, const double* d, double t_min, double* t, int* idx, double* p,
              double* nrm, int* front) {
    const Scene& sc = *static_cast<Scene*>(scene);
    for (int k = 0; k < n; ++k) {
        Ray r{{o[3 * k], o[3 * k + 1], o[3 * k + 2]}, {d[3 * k], d[3 * k + 1], d[3 * k + 2]}};
        Hit h;
        bool hit = list_hit(sc, r, t_min, std::numeric_limits<double>::infinity(), h);
        idx[k] = hit ? h.idx : -1;
        t[k] = hit ? h.t : 0;
        front[k] = hit && h.front;
        p[3 * k] = h.p.x; p[3 * k + 1] = h.p.y; p[3 * k + 2] = h.p.z;
        nrm[3 * k] = h.n.x; nrm[3 * k + 1] = h.n.y; nrm[3 * k + 2] = h.n.z;
    }
}

// T4: scatter with injected uniforms. u is [n][3]; d_in, p, nrm, d_out, atten are [n][3].
void orc_scatter(int n, const int* kind, const double* albedo, const double* fuzz, const double* ior,
                 const double* d_in, const double* p, const double* nrm, const int* front, const double* u,
                 double* d_out, double* atten, int* ok) {
    for (int k = 0; k < n; ++k) {
        SphereRec m{{0, 0, 0}, 1, kind[k], {albedo[3 * k], albedo[3 * k + 1], albedo[3 * k + 2]}, fuzz[k], ior[k]};
        Hit h;
        h.p = {p[3 * k], p[3 * k + 1], p[3 * k + 2]};
        h.n = {nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2]};
        h.front = front[k] != 0;
        Ray in{{0, 0, 0}, {d_in[3 * k], d_in[3 * k + 1], d_in[3 * k + 2]}};
        V3 a; Ray out;
        ok[k] = scatter(m, in, h, u[3 * k], u[3 * k + 1], u[3 * k + 2], a, out);
        d_out[3 * k] = out.d.x; d_out[3 * k + 1] = out.d.y; d_out[3 * k + 2] = out.d.z;
        atten[3 * k] = a.x; atten[3 * k + 1] = a.y; atten[3 * k + 2] = a.z;
    }
}

// Radiance of given rays with philox-numbered uniforms (pixel, sample per ray).
void orc_sky(int n, const double* d, double* rgb) {
    for (int k = 0; k < n; ++k) {
        V3 ud = unit({d[3 * k], d[3 * k + 1], d[3 * k + 2]});
        double t = 0.5 * (ud.y + 1);
        rgb[3 * k] = (1.0 - t) + 0.5 * t; rgb[3 * k + 1] = (1.0 - t) + 0.7 * t; rgb[3 * k + 2] = (1.0 - t) + 1.0 * t;
    }
}

// calcColour (Path Tracer.cpp:82-108) for given rays with the uniform numbers of every scatter injected: u is
// [n][max_depth][3], the scatter at the k-th hit of path i takes u[i][k] = (u1, u2 for the sphere point, u3 for the
// Fresnel coin). rgb is [n][3]; n_rays[i] = calcColour entries with depth > 0 (rays traced) of path i.
void orc_trace(void* scene, int n, const double* o, const double* d, int max_depth, const double* u, double* rgb, int* n_rays) {
    const Scene& sc = *static_cast<Scene*>(scene);
    for (int k = 0; k < n; ++k) {
        Ray r{{o[3 * k], o[3 * k + 1], o[3 * k + 2]}, {d[3 * k], d[3 * k + 1], d[3 * k + 2]}};
        Sampler sm;
        sm.mode = 2;
        sm.inject = u + size_t(k) * size_t(max_depth > 0 ? max_depth : 0) * 3;
        uint64_t rays = 0;
        V3 c = radiance(sc, r, max_depth, 0, sm, rays);
        rgb[3 * k] = c.x; rgb[3 * k + 1] = c.y; rgb[3 * k + 2] = c.z;
        n_rays[k] = int(rays);
    }
}

// The render. Adds samples [spp_begin, spp_end) of every pixel into sum (and their squares into
// sumsq when given), top row first like the reference's pixel vector (Path Tracer.cpp:298).
//   mode 0: Mersenne-Twister streams as the reference draws them. Worker k seeds RNG-A with
//           seedA + k (the reference seeds it from random_device per thread, :49).
//           shared_b = 1: one RNG-B/RNG-C shared by all workers under a mutex (VectorFunc.cpp:18;
//           RNG-C is an unguarded race in the reference and is put under the same lock here).
//           shared_b = 0: per-worker RNG-B/RNG-C, no lock (faster, same distribution).
//   mode 1: philox uniforms keyed (seed_lo, seed_hi), numbered (pixel = j*W + i, sample, bounce).
// Threads take rows from a shared counter in submission order (the reference queues one task
// per row and hardware_concurrency() workers pop them, ThreadPool.h:119-138, Path Tracer.cpp:303-305).
void orc_render(void* scene, const double* cam19, int W, int H, int spp_begin, int spp_end, int max_depth, int mode,
                uint32_t seedA, uint32_t seedB, int threads, int shared_b, double* sum, double* sumsq,
                uint64_t* rays_out) {
    const Scene& sc = *static_cast<Scene*>(scene);
    Cam cam = cam_from(cam19);
    if (threads < 1) threads = 1;
    std::atomic<int> next_row{0};
    std::atomic<uint64_t> total_rays{0};
    std::mutex lock;
    MT19937 sharedB(seedB), sharedC(5489u);
    auto worker = [&](int k) {
        MT19937 ownA(seedA + uint32_t(k)), ownB(seedB + uint32_t(k)), ownC(5489u + uint32_t(k));
        Sampler sm;
        sm.mode = mode;
        sm.a = &ownA;
        sm.b = shared_b ? &sharedB : &ownB;
        sm.c = shared_b ? &sharedC : &ownC;
        sm.lock = (mode == 0 && shared_b) ? &lock : nullptr;
        sm.key[0] = seedA; sm.key[1] = seedB;
        uint64_t rays = 0;
        for (;;) {
            int j = next_row.fetch_add(1);
            if (j >= H) break;
            for (int i = 0; i < W; ++i) {
                size_t at = (size_t(W) * size_t(H - j - 1) + size_t(i)) * 3;      // :298 row flip
                sm.pixel = uint32_t(j) * uint32_t(W) + uint32_t(i);
                for (int s = spp_begin; s < spp_end; ++s) {                         // :267
                    sm.sample = uint32_t(s);
                    V3 c = sample_pixel(sc, cam, W, H, i, j, max_depth, sm, rays);
                    sum[at] += c.x; sum[at + 1] += c.y; sum[at + 2] += c.z;         // :275
                    if (sumsq) { sumsq[at] += c.x * c.x; sumsq[at + 1] += c.y * c.y; sumsq[at + 2] += c.z * c.z; }
                }
            }
        }
        total_rays += rays;
    };
    std::vector<std::thread> pool;
    for (int k = 1; k < threads; ++k) pool.emplace_back(worker, k);
    worker(0);
    for (auto& t : pool) t.join();
    if (rays_out) *rays_out = total_rays.load();
}

// T7: the P3 file the reference writes (Path Tracer.cpp:316-322). Returns the byte count, or the
// needed capacity (negated) when cap is too small.
long orc_format_ppm(const double* sum, int W, int H, int spp, char* out, long cap) {
    std::string s = "P3\n" + std::to_string(W) + " " + std::to_string(H) + "\n255\n";
    for (size_t k = 0; k < size_t(W) * size_t(H); ++k) {
        s += std::to_string(quantise(sum[3 * k], spp)) + " " + std::to_string(quantise(sum[3 * k + 1], spp)) + " " +
             std::to_string(quantise(sum[3 * k + 2], spp)) + "\n";
    }
    if (long(s.size()) > cap) return -long(s.size());
    std::memcpy(out, s.data(), s.size());
    return long(s.size());
}

}   // extern "C"

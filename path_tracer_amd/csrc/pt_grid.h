// pt_grid.h — optional closest-hit search through a uniform grid, for scenes
// where the linear scan of HittableList.cpp:27 over every sphere dominates (10k spheres and more).
//
// The answer is the one HittableList::isHit gives (closest t in range, ties to the later list index,
// HittableList.cpp:28-31 with Sphere.cpp:18); only the set of spheres looked at shrinks:
//   big list   spheres that would cover many cells (the ground sphere of Path Tracer.cpp:183) stay in a
//              short list scanned with the scalar-register scan of pt_device.h;
//   grid       every other sphere is entered in each cell its box touches. A lane walks the cells its
//              ray crosses (3-D DDA) and tests the spheres entered there, relative to the ray origin:
//              oc = c - o, b = oc.dh, T = r^2 - |oc - b dh|^2, roots b -+ sqrt(T) (Sphere.cpp:6-19 in
//              unit-direction form, the discriminant written so that it keeps its digits at any
//              distance). The walk ends when the closest hit so far lies before the exit of the current
//              cell, or the ray leaves the grid.
// Lanes diverge here by design (one path per lane, one cell list per lane): loads are per-lane vector
// loads that hit in L2 (the whole structure is a few hundred KB), not scalar broadcasts.
#pragma once

#include "pt_device.h"

namespace ptd {

struct GridDev {
    const uint32_t* cell_start;   // nx*ny*nz + 1 offsets into items / ids
    const float4* items;          // per entry: cx, cy, cz, r^2 of the sphere (floats of the double scene)
    const int* ids;               // per entry: list index of the sphere
    const int* big_ids;           // list index of each sphere of the big list
    const int* big_local;         // per list index: position in the big list, or -1
    SceneDev big;                 // scan records of the big list (geom / mat unused)
    int last_cell;                // nx * ny * nz - 1
    const float* shape;           // device copy of a GridShape: read with scalar loads where a walk is set up
};
struct GridShape {                // 11 dwords
    float gmin[3], gmax[3];       // box the cells tile
    float cell, inv_cell;         // cell edge
    int nx, ny, nz;
};
typedef const float __attribute__((address_space(4)))* grid_const_float_ptr;
typedef const int __attribute__((address_space(4)))* grid_const_int_ptr;

PT_DEV float nonzero(float d) { return __builtin_fabsf(d) < 1e-12f ? __builtin_copysignf(1e-12f, d) : d; }

// A walk in progress, per lane.
struct GridWalk {
    float tmx, tmy, tmz;          // distance along the ray to the next cell wall on each axis
    float tdx, tdy, tdz;          // distance between walls
    float t_out;                  // far side of the box
    float best_t;                 // closest hit so far (starts as the big list's)
    int best_i;
    int ci, sx, sy, sz;           // cell index and its step on each axis
    uint32_t k, kend;             // entries of the current cell still to test
    bool prev_in_grid;            // the sphere the ray starts on lives in the grid (else the big list handled it)
    bool live;
};

// Starts the search for one ray: scans the big list (scalar-register scan; its indices are positions in that
// list), clips the ray to the box of the grid and sets up the walk. o, dh, tmin, prev, self_t as scan_spheres.
PT_DEV void grid_begin(const GridDev& G, f3 o, f3 dh, float tmin, int prev, float self_t, GridWalk& w) {
    int prev_local = -1;
    if (prev >= 0) prev_local = G.big_local[prev];
    float best_t; int best_i;
    scan_spheres(G.big, o, dh, tmin, prev_local, self_t, best_t, best_i);
    if (best_i >= 0) best_i = G.big_ids[best_i];
    w.best_t = best_t; w.best_i = best_i;
    w.prev_in_grid = (prev >= 0) & (prev_local < 0);

    // The shape of the grid comes through the scalar cache here, where it is used (an opaque copy of the pointer
    // keeps hipcc from hoisting the loads and parking eleven more values in vector lanes for the whole kernel).
    const float* shape = G.shape;
    asm volatile("" : "+s"(shape));
    grid_const_float_ptr gf = (grid_const_float_ptr)(uintptr_t)shape;
    grid_const_int_ptr gi = (grid_const_int_ptr)(uintptr_t)shape;
    const f3 gmin = mk3(gf[0], gf[1], gf[2]), gmax = mk3(gf[3], gf[4], gf[5]);
    const float cell = gf[6], inv_cell = gf[7];
    const int nx = gi[8], ny = gi[9], nz = gi[10];

    // ---- clip the ray to the box of the grid ----
    const f3 dn = mk3(nonzero(dh.x), nonzero(dh.y), nonzero(dh.z));   // the walk only: a zero component would make 0 * inf
    const f3 inv = mk3(fast_rcp(dn.x), fast_rcp(dn.y), fast_rcp(dn.z));
    const f3 ta = (gmin - o) * inv, tb = (gmax - o) * inv;
    const float t_in = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ta.x, tb.x), __builtin_fminf(ta.y, tb.y)),
                                       __builtin_fmaxf(__builtin_fminf(ta.z, tb.z), 0.0f));
    w.t_out = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ta.x, tb.x), __builtin_fmaxf(ta.y, tb.y)),
                              __builtin_fmaxf(ta.z, tb.z));   // far side of the box
    w.live = (t_in <= w.t_out) & (t_in <= best_t);

    // ---- the cell the ray enters at, the distance to the next cell wall on each axis ----
    const f3 q = (fma3(dh, t_in, o) - gmin) * inv_cell;
    const int ix = min(max(int(__builtin_floorf(q.x)), 0), nx - 1);
    const int iy = min(max(int(__builtin_floorf(q.y)), 0), ny - 1);
    const int iz = min(max(int(__builtin_floorf(q.z)), 0), nz - 1);
    const bool fx = dn.x > 0.0f, fy = dn.y > 0.0f, fz = dn.z > 0.0f;
    w.tmx = (__builtin_fmaf(float(ix + (fx ? 1 : 0)), cell, gmin.x) - o.x) * inv.x;
    w.tmy = (__builtin_fmaf(float(iy + (fy ? 1 : 0)), cell, gmin.y) - o.y) * inv.y;
    w.tmz = (__builtin_fmaf(float(iz + (fz ? 1 : 0)), cell, gmin.z) - o.z) * inv.z;
    w.tdx = cell * __builtin_fabsf(inv.x); w.tdy = cell * __builtin_fabsf(inv.y); w.tdz = cell * __builtin_fabsf(inv.z);
    w.sx = fx ? 1 : -1; w.sy = fy ? nx : -nx; w.sz = fz ? nx * ny : -(nx * ny);
    w.ci = (iz * ny + iy) * nx + ix;
    w.k = 0; w.kend = 0;
    if (w.live) { w.k = G.cell_start[w.ci]; w.kend = G.cell_start[w.ci + 1]; }
}

// One pass of a live walk: either tests one entry of the current cell or moves to the next cell, chosen with
// selects so that the lanes of a wave run one instruction stream. The entry of this pass and the cell range
// of the next are requested together: one memory round trip per pass.
// A lane leaves (live = false) when its closest hit lies before the exit of its cell, or at the far side of
// the box (should rounding let it take one more step, the cell index is clamped to the table: a cell too many
// is looked at, which cannot change the answer, and the next exit distance lies beyond the box for certain).
// Every pass consumes an entry or moves one cell on: at most nx + ny + nz + 3 + entries passes per walk.
// FLAT: the grid is one layer of cells (ny = 1, the reference's scenes: spheres on the ground). The walk then
// never moves along y; where the ray leaves the layer is the far side of the box, t_out.
// tmin: the ray's, as given to grid_begin (the caller recomputes it from the ray's length rather than keep it in a register).
template <bool FLAT>
PT_DEV void grid_pass(const GridDev& G, f3 o, f3 dh, float tmin, int prev, GridWalk& w) {
    const bool has = w.k < w.kend;
    const float4 s = G.items[w.k];                    // one entry past the last is allocated: k == kend is readable
    const int id = G.ids[w.k];
    // -- the step, for the lanes whose cell is used up --
    float t_exit;
    bool stop;
    int ci;
    if constexpr (FLAT) {
        t_exit = __builtin_fminf(w.tmx, w.tmz);
        stop = !has & ((w.best_t < t_exit) | (t_exit >= w.t_out));
        const bool st = !has & !stop;
        const bool ax = w.tmx <= w.tmz;
        const bool mx = st & ax, mz = st & !ax;
        w.tmx += mx ? w.tdx : 0.0f;
        w.tmz += mz ? w.tdz : 0.0f;
        ci = w.ci + (mx ? w.sx : (mz ? w.sz : 0));
    } else {
        t_exit = __builtin_fminf(__builtin_fminf(w.tmx, w.tmy), w.tmz);
        stop = !has & ((w.best_t < t_exit) | (t_exit >= w.t_out));
        const bool st = !has & !stop;
        const bool ax = (w.tmx <= w.tmy) & (w.tmx <= w.tmz), ay = !ax & (w.tmy <= w.tmz);
        const bool mx = st & ax, my = st & ay, mz = st & !ax & !ay;
        w.tmx += mx ? w.tdx : 0.0f;
        w.tmy += my ? w.tdy : 0.0f;
        w.tmz += mz ? w.tdz : 0.0f;
        ci = w.ci + (mx ? w.sx : (my ? w.sy : (mz ? w.sz : 0)));
    }
    ci = min(max(ci, 0), G.last_cell);
    w.ci = ci;
    const uint32_t r0 = G.cell_start[ci], r1 = G.cell_start[ci + 1];
    // -- the test, for the lanes with an entry: Sphere.cpp:6-19 relative to the ray origin --
    const f3 oc = mk3(s.x - o.x, s.y - o.y, s.z - o.z);                  // Sphere.cpp:6 (sign folded into b)
    const float b = dot(oc, dh);                                          // :8
    // disc = r^2 - (distance from the centre to the ray's line)^2, from the perpendicular part of oc:
    // b*b - |oc|^2 cancels badly once the sphere is far away compared with its radius
    const f3 perp = fma3(dh, -b, oc);
    const float T = __builtin_fmaf(-perp.z, perp.z, __builtin_fmaf(-perp.y, perp.y, __builtin_fmaf(-perp.x, perp.x, s.w)));   // :9-10
    const float sq = fast_sqrt(T);                                        // NaN for T < 0: t is NaN, the compares below false
    const float t1 = b - sq, t2 = b + sq;
    const float t = (t1 >= tmin) ? t1 : t2;                               // :17-19
    const bool closer = (t < w.best_t) | ((t == w.best_t) & (id > w.best_i));   // HittableList.cpp:28-31: later index wins ties
    const bool ok = has & (id != prev) & (t >= tmin) & closer;
    w.best_t = ok ? t : w.best_t;
    w.best_i = ok ? id : w.best_i;
    w.k = has ? w.k + 1 : r0;
    w.kend = has ? w.kend : r1;
    w.live = !stop;
}

// The answer of a finished walk. The sphere the ray starts on, when it lives in the grid: its near root is the
// start point, its far root (self_t) counts when the ray heads into the sphere (see scan_spheres).
PT_DEV void grid_end(const GridWalk& w, float tmin, int prev, float self_t, float& best_t, int& best_i) {
    const bool ok = w.prev_in_grid & (self_t >= tmin) & (self_t <= w.best_t);
    best_t = ok ? self_t : w.best_t;
    best_i = ok ? prev : w.best_i;
}

// Same contract as scan_spheres (pt_device.h): best_i is a LIST index. The whole search for one ray per lane;
// the wave stays in the loop until its last lane is done (the sample loop interleaves walks instead).
template <bool FLAT>
PT_DEV void scan_grid(const GridDev& G, f3 o, f3 dh, float tmin, int prev, float self_t, float& best_t, int& best_i) {
    GridWalk w;
    grid_begin(G, o, dh, tmin, prev, self_t, w);
    while (__builtin_amdgcn_ballot_w64(w.live)) {
#ifdef PT_STATS
        {   // passes of the wave, lanes still walking, lanes with an entry to test
            const unsigned long long lm = __builtin_amdgcn_ballot_w64(w.live), hm = __builtin_amdgcn_ballot_w64(w.live && w.k < w.kend);
            if (threadIdx.x == 0) { atomicAdd(&g_stats[5], 1ull); atomicAdd(&g_stats[6], (unsigned long long)__popcll(lm)); atomicAdd(&g_stats[7], (unsigned long long)__popcll(hm)); }
        }
#endif
        if (w.live) grid_pass<FLAT>(G, o, dh, tmin, prev, w);
    }
    grid_end(w, tmin, prev, self_t, best_t, best_i);
}

}   // namespace ptd

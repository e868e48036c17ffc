// pt_hip.hip — kernels and C-ABI of libpt_hip.so (gfx950 only; see include/pt_hip.h).
//
// Kernels
//   k_render<MODE>  the sample loop (MODE 0: linear scan of every sphere, the default; MODE 1 and 2: the same
//              loop with the closest-hit search through the uniform grid of pt_grid.h): per-lane iterative bounce loop with path regeneration, persistent
//              waves. A wave (one 64-thread workgroup) pulls jobs = an 8x8 pixel tile x a chunk of
//              samples from a global counter and hands (pixel, sample) items to its lanes as their
//              paths end, so every lane stays inside the sphere scan. Replaces process_row /
//              sumColour / calcColour (Path Tracer.cpp:296-300, :265-278, :82-108).
//   k_resolve  converts the job sums (64-bit fixed point, order-independent) to float and adds
//              them into the caller's accumulation buffer.
//   k_scan / k_scatter / k_gen_ray / k_sky / k_philox   the same device functions, one thread
//              per element, behind the unit-test entry points.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/pt_hip.h"
#include "pt_device.h"
#include "pt_grid.h"

using namespace ptd;

// =============================================================================================
// Device side
// =============================================================================================
// RenderParams::counters in u64 words: [0] rays of the call, [3] rays since the handle was made (the handle's own block
// only), [8 + 16 k] job counter k of PT_JOB_COUNTERS (one cache line each), then the hand-over slots.
#ifndef PT_JOB_COUNTERS
#define PT_JOB_COUNTERS 2           // a power of two (measured: 2, 4 and 8 serve the 400x225 frame alike, 2 costs configs[1] least)
#endif
#define PT_COUNTER_WORDS (8 + 16 * PT_JOB_COUNTERS)
#define PT_MAX_SPP_CHUNK 65536      // samples per pixel in one job: 8 rows x 65,536 = 2^19 row-items, below the 2^20 of the reciprocal trick in k_render
struct TailPath {                    // a path handed from k_render to k_tail: the ray it is about to trace and what it carries
    float o[3], dh[3], thr[3];
    float len, self_t;
    int prev, bounce;                // bounce < 0: empty slot
    uint32_t pixel, sample, top_pixel;
};
static_assert(sizeof(TailPath) == 64, "TailPath is 16 words");

struct RenderParams {
    int W, H;
    // Jobs, in the order the counter hands them out: n_coarse layers of (every tile x spp_chunk samples per pixel), then
    // layers of spp_fine samples per pixel from sample fine_begin on. Waves run at very different speeds (the oldest wave
    // of a SIMD issues first) and take jobs as they go, so the frame ends within about the length of the LAST jobs:
    // those are small, the bulk is large (fewer trips to the counter and to the frame).
    int spp_begin, spp_end, spp_chunk, n_coarse, spp_fine, fine_begin, n_tiles;
    int tiles_x;
    int n_jobs;
    int max_depth;
    uint32_t key0, key1;
    float inv_w1, inv_h1;
    const float* cam_dev;            // the 19 floats of a CameraDev in device memory (read with scalar loads where used)
    SceneDev scene;
    GridDev grid;                    // read by k_render<1> and k_render<2> only
    unsigned long long* accum_fx;    // [3][H][W] fixed-point (2^-32) sums, top row first
    // One block of device memory: PT_COUNTER_WORDS u64 of counters, then the slots through which k_render<0> hands paths
    // to k_tail (PT_TAIL_PATHS slots per wave of the launch). Words: [0] rays of this call; [3] rays since the handle was
    // made (the handle's own block only); [8 + 16 k] job counter k of PT_JOB_COUNTERS, a u32 on a cache line of its own
    // (the launch sets it to the jobs its waves took as their first, whose numbers are the waves' own). One pointer
    // instead of three: the scan leaves no scalar register to spare.
    unsigned long long* counters;
    int n_tail_slots;                // k_tail only: slots written by the k_render launch before it
};
PT_DEV unsigned long long* ray_counter(const RenderParams& P) { return P.counters; }
PT_DEV unsigned int* job_counter(const RenderParams& P, unsigned k) { return reinterpret_cast<unsigned int*>(P.counters + 8 + 16 * k); }
PT_DEV TailPath* tail_slots(const RenderParams& P) { return reinterpret_cast<TailPath*>(P.counters + PT_COUNTER_WORDS); }

// A CameraDev from its 19 floats in device memory; the constant address space makes hipcc use scalar loads
// although the kernel also contains atomics.
typedef const float __attribute__((address_space(4)))* const_float_ptr;
PT_DEV CameraDev load_camera(const float* p) {
    const_float_ptr f = (const_float_ptr)(uintptr_t)p;
    CameraDev c;
    c.pos = mk3(f[0], f[1], f[2]); c.ll_rel = mk3(f[3], f[4], f[5]); c.hor = mk3(f[6], f[7], f[8]);
    c.ver = mk3(f[9], f[10], f[11]); c.u = mk3(f[12], f[13], f[14]); c.v = mk3(f[15], f[16], f[17]);
    c.lens = f[18];
    return c;
}

// Radiance per channel to unsigned Q32.32. It is never negative and never a NaN: the sky is in [0.5, 1] and every
// attenuation is a finite, non-negative albedo or 1 (pt_set_scene rejects anything else), so no clamp is needed here
// (round 3: two instructions per channel and path fewer); the conversions saturate what exceeds 2^32.
// Integer sums make the frame independent of the order in which lanes and jobs finish.
PT_DEV unsigned long long to_fx(float v) {
    const uint32_t hi = uint32_t(v);                                 // whole part: the conversion truncates and saturates
    const uint32_t lo = uint32_t(__builtin_amdgcn_fractf(v) * 4294967296.0f);   // fraction, exact (a power-of-two scale), truncated
    return ((unsigned long long)hi << 32) | lo;
}

#ifdef PT_STATS
// Diagnostic build only (tools/stats.sh): wave-level counts of where the instruction stream goes.
__device__ unsigned long long g_stats[16];
#define PT_STAT_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
#else
#define PT_STAT_ADD(i, v) do { } while (0)
#endif

// Adds one LDS bank of job sums to the frame and clears it. Lane l owns pixel slot l of the tile.
PT_DEV void flush_bank(const RenderParams& P, unsigned long long* bank, int tx, int ty, int lane) {
    // Opaque copy: everything derived from the lane number below (the LDS offset, the pixel) is then computed here, at
    // a job switch, instead of being kept in registers through the whole loop (the grid kernels have none to spare
    // and would spill them).
    asm volatile("" : "+v"(lane));
    const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    if (px < P.W && py < P.H) {
        // planar sums: the 8 lanes of a tile row add to 64 contiguous bytes
        const size_t plane = size_t(P.W) * size_t(P.H);
        unsigned long long* dst = P.accum_fx + size_t(py) * size_t(P.W) + size_t(px);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned long long v = bank[lane * 3 + c];
            if (v) atomicAdd(dst + c * plane, v);
        }
    }
    unsigned long long zero = 0;
    asm volatile("" : "+v"(zero));
    bank[lane * 3 + 0] = zero; bank[lane * 3 + 1] = zero; bank[lane * 3 + 2] = zero;
}

#ifdef PT_WAVEPROF
#ifndef PT_SECPROF_AFTER
#define PT_SECPROF_AFTER 10000     // count the bounces that start this long (10 ns units) after the wave did: the others share their SIMD with busy waves
#endif
__device__ unsigned long long g_tprof2[8192 * 2];   // per wave of k_tail: 10 ns units spent in iterations with one path, with several
__device__ unsigned long long g_sprof[16];     // sections of the bounces a path has a wave to itself: [0] bounces, [1] scan, [2] finish, [3] scatter (10 ns units)
#endif
// Finishes the paths in slots[0 .. n_slots) (bounce < 0: empty slot), a path per GROUP of G lanes: the lanes of a group hold
// the same path and share out the spheres (coop_scan); same arithmetic per sphere, same hit rule, same philox blocks and
// integer sums straight to the frame as the sample loop, so the frame is bit for bit what the loop alone would produce. The
// groups of the calling waves take the slots one by one through *next (a counter in LDS, zero on entry): a path of fifty
// bounces then holds up one group while the others work through the rest. Returns this lane's count of rays traced.
// Used by k_tail (slots in global memory written by k_render, scene copies in LDS) and by the end of k_render itself.
template <int G, class Quads>
PT_DEV unsigned int finish_paths(const RenderParams& P, const TailPath* slots, int n_slots, int* next, const Quads& quads,
                                 const GeomRec* geom, const MatRec* mats, int tid, unsigned long long* iterations = nullptr) {
    const int sub = tid & (G - 1);
    bool more = true;                                                     // this group has not yet seen the end of the run
    unsigned int rays = 0;
    bool active = false;
    f3 o = mk3(0, 0, 0), dh = mk3(0, 0, 1), thr = mk3(0, 0, 0);
    float len = 1.0f, self_t = 0.0f;
    int prev = -1, bounce = 0;
    uint32_t pixel = 0, sample = 0, top_pixel = 0;
    u4 pc; pc.x = pc.y = pc.z = pc.w = 0u;                                // philox block of bounce pc_first + sub of the group's path
    int pc_first = -(1 << 30);
#ifdef PT_WAVEPROF
    __shared__ unsigned long long sec_lds[4][12];                         // this wave's sums (one lane adds to them; registers would cost the build a third of its waves)
    unsigned long long* const sec = sec_lds[(tid >> 6) & 3];
    if ((tid & 63) < 12) sec[tid & 63] = 0;
    const uint32_t sec_start = uint32_t(wall_clock64());
#endif
    auto take_slot = [&](int take) {
        const TailPath tp = slots[take];
        if (tp.bounce < 0) return;
        o = mk3(tp.o[0], tp.o[1], tp.o[2]); dh = mk3(tp.dh[0], tp.dh[1], tp.dh[2]); thr = mk3(tp.thr[0], tp.thr[1], tp.thr[2]);
        len = tp.len; self_t = tp.self_t; prev = tp.prev; bounce = tp.bounce;
        pixel = tp.pixel; sample = tp.sample; top_pixel = tp.top_pixel;
        pc_first = -(1 << 30);                                            // the philox blocks held are another path's
        active = true;
    };
    for (;;) {
        // a group without a path takes the next occupied slot (the lanes of a group read the same words)
        while (!active && more) {
            int take = 0;
            if (sub == 0) take = atomicAdd(next, 1);
            take = __shfl(take, tid & 63 & ~(G - 1));
            if (take >= n_slots) { more = false; break; }
            take_slot(take);
        }
        const unsigned long long with_path = __ballot(active);
        if (with_path == 0ull) break;                                     // (waves leave independently: no barrier in here)
        if (iterations) *iterations += 1ull + ((G < 64 && __popcll(with_path) == G) ? (1ull << 20) : 0ull);   // (the per-wave timing build only; alone-mode iterations in the upper bits)
        // ONE path left in this wave (the end of a frame is a few long paths, each alone in its wave): all 64 lanes share
        // out its spheres, a quarter of the quads per lane and of the round trips to the records. The other lanes fetch
        // the ray from the group that holds it; the answer is the same in every lane (same arithmetic per sphere, same rule).
        float bt = 0.0f; int bi = -1;
        const int n_paths = __popcll(with_path) / G;
        const bool alone = G < 64 && n_paths == 1;
#ifdef PT_WAVEPROF
        const uint32_t sp0 = uint32_t(wall_clock64());                  // (32 bits: the differences are what is used)
        uint32_t sp1 = sp0, sp2 = sp0;
#endif
        if (alone) {                                                      // wave-uniform; every lane takes part
            const int holder = __ffsll(with_path) - 1;                    // wave-uniform: the ray comes through scalar registers
            auto from_holder = [holder](float v) { return __uint_as_float(uint32_t(__builtin_amdgcn_readlane(int(__float_as_uint(v)), holder))); };
            const f3 so = mk3(from_holder(o.x), from_holder(o.y), from_holder(o.z));
            const f3 sd = mk3(from_holder(dh.x), from_holder(dh.y), from_holder(dh.z));
            const float sl = from_holder(len), ss = from_holder(self_t);
            const int sp = __builtin_amdgcn_readlane(prev, holder);
            coop_scan<64>(quads, P.scene.n_pad, tid & 63, so, sd, 0.001f * sl, sp, ss, bt, bi);
        }
        if (active) {
            if (sub == 0) ++rays;
            if (!alone) coop_scan<G>(quads, P.scene.n_pad, sub, o, dh, 0.001f * len, prev, self_t, bt, bi);
#ifdef PT_WAVEPROF
            asm volatile("" :: "v"(bt), "v"(bi));
            sp1 = sp2 = uint32_t(wall_clock64());
#endif
            if (bi < 0) {                                                            // Path Tracer.cpp:105-107
                const f3 c = thr * sky(dh.y);
                if (sub == 0) {
                    const size_t plane = size_t(P.W) * size_t(P.H);
                    unsigned long long* dst = P.accum_fx + top_pixel;
                    const unsigned long long vx = to_fx(c.x), vy = to_fx(c.y), vz = to_fx(c.z);
                    if (vx) atomicAdd(dst, vx);
                    if (vy) atomicAdd(dst + plane, vy);
                    if (vz) atomicAdd(dst + 2 * plane, vz);
                }
                active = false;
            } else {
                const GeomRec g = geom[bi];
                const MatRec mat = mats[bi];
                const HitRec h = finish_hit(g, bi == prev, o, dh, 0.001 * double(len));
#ifdef PT_WAVEPROF
                asm volatile("" :: "v"(h.p.x), "v"(h.n.x), "v"(h.n.z));
                sp2 = uint32_t(wall_clock64());
#endif
                prev = bi;
                ++bounce;
                if (bounce >= P.max_depth) {                                         // Path Tracer.cpp:86
                    active = false;
                } else {
                    // The philox block of this scatter. A block depends on (pixel, sample, bounce) only, so the G lanes of a
                    // group, instead of all computing the same one, compute those of the next G bounces at once (lane `sub`:
                    // bounce pc_first + sub) and every scatter fetches its own from the lane that holds it: the block's ten
                    // dependent rounds leave the per-bounce chain of a long path except once in G bounces. When one group of
                    // the wave needs new blocks all groups in this branch take theirs (same instructions either way).
                    if (__any(bounce < pc_first || bounce >= pc_first + G)) {
                        pc_first = bounce;
                        pc = philox4x32_10(pixel, sample, uint32_t(bounce + sub), 0u, P.key0, P.key1);
                    }
                    const int holder = (tid & 63 & ~(G - 1)) + (bounce - pc_first);
                    const uint32_t rx = __shfl(pc.x, holder), ry = __shfl(pc.y, holder), rz = __shfl(pc.z, holder);
                    f3 d, atten;
                    const bool ok = shade(mat, dh, len, h.n, h.front, u24(rx), u24(ry), u24(rz), d, atten);
                    thr = thr * atten;                                               // Path Tracer.cpp:97
                    o = h.p;
                    dh = normalise(d, len);
                    self_t = self_far_root(h.r, h.n, h.front, dh);
                    active = ok;                                                     // Path Tracer.cpp:100 -> black
                }
            }
#ifdef PT_WAVEPROF
            if (sub == 0) {
                asm volatile("" :: "v"(dh.x), "v"(self_t), "v"(len));
                const uint32_t sp3 = uint32_t(wall_clock64());
                if ((tid & 63) == __ffsll(with_path) - 1) sec[alone ? 9 : 10] += sp3 - sp0;        // every iteration of the wave, by kind
                if (sp0 - sec_start > PT_SECPROF_AFTER) {
                    if (alone) { sec[0] += 1; sec[1] += sp1 - sp0; sec[2] += sp2 - sp1; sec[3] += sp3 - sp2; }
                    else if ((tid & 63) == __ffsll(with_path) - 1) { sec[4] += 1; sec[5] += sp1 - sp0; sec[6] += sp2 - sp1; sec[7] += sp3 - sp2; sec[8] += __popcll(with_path) / G; }
                }
            }
#endif
        }
    }
#ifdef PT_WAVEPROF
    if ((tid & 63) < 9 && sec[tid & 63]) atomicAdd(&g_sprof[tid & 63], sec[tid & 63]);
    {
        const unsigned w = blockIdx.x * 4 + (tid >> 6);
        if ((tid & 63) < 2 && w < 8192) g_tprof2[2 * w + (tid & 63)] = sec[9 + (tid & 63)];
    }
#endif
    return rays;
}

// Persistent waves. Each wave pulls jobs (8x8 tile x sample chunk) from a global counter and hands
// the job's (sample, pixel-slot) items to its lanes as their paths end. When a job's items run out
// the wave moves straight on to the next job while the last paths of the previous one are still in
// flight (two LDS banks of sums; the older one is flushed at the switch and its stragglers then add to the
// frame directly), so lanes idle only at the very end of the frame.
#ifndef PT_WAVES_PER_EU
#define PT_WAVES_PER_EU 6
#endif
#ifndef PT_GRID_WAVES_PER_EU
#define PT_GRID_WAVES_PER_EU 5      // 96 registers, no spills; at 8 waves (64) the walk state spills and the kernel runs at half the speed
#endif
#ifndef PT_PRIO
#define PT_PRIO 1
#endif
#ifndef PT_TAIL_PATHS
#define PT_TAIL_PATHS 16            // a wave left with at most this many paths and no job hands them to k_tail
#endif
#ifndef PT_TAIL_GROUP
#define PT_TAIL_GROUP 16            // lanes that share one path at the end of a frame
#endif
#ifndef PT_GRID_RATIO
#define PT_GRID_RATIO 3             // answers are handled once they outnumber the walks in progress by this factor
#endif
#ifdef PT_WAVEPROF
__device__ unsigned long long g_rprof[8192 * 4 + 8];
#endif
template <int MODE>
__global__ void __launch_bounds__(64, MODE ? PT_GRID_WAVES_PER_EU : PT_WAVES_PER_EU) k_render(const RenderParams P) {
    __shared__ unsigned long long acc[2 * 192];
#ifdef PT_WAVEPROF
    __shared__ unsigned long long rp[3];
    if (threadIdx.x == 0) { rp[0] = wall_clock64(); rp[1] = 0; rp[2] = 0; }
#endif
    __shared__ unsigned int wave_number;            // blockIdx.x, parked in LDS until the end of the kernel: the loop has no scalar register for it
    const int lane = threadIdx.x;
#if PT_PRIO
    // The oldest wave of a SIMD issues first, and in this kernel the vector ALU is always wanted: the youngest of the six
    // runs at a tenth of the speed of the oldest (measured: rays per wave 12,000 to 168,000 in one launch) and would still
    // be in the middle of a large job when every other wave has left. A wave on a large job therefore runs at a higher
    // priority than one on the small jobs of the end, and that one higher than a wave that only drains its last paths.
    __builtin_amdgcn_s_setprio(2);
#endif
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[lane + 64 * i] = 0;
    __shared__ unsigned int my_counter;             // the job counter this wave asks first (it moves on when that one has run out)
    if (lane == 0) { wave_number = blockIdx.x; my_counter = blockIdx.x % PT_JOB_COUNTERS; }

    // wave-uniform job state: `c_` the job items are drawn from, `o_` the one still draining. Tile, first sample and row
    // count of the two jobs live in LDS, not in registers: job_lds[parity] = the current job's (tx, ty, s0, vh),
    // job_lds[parity ^ 1] the draining one's, like the two banks of sums. The scan leaves no scalar register for them, and
    // as loop-carried vector registers they cost copies in every iteration although only the regeneration below reads
    // them and only a job switch writes them. A new job is written to the slot of the job that has just been flushed;
    // a lane then reads the record of the job its item comes from with ONE ds_read_b128 at a per-lane address.
    __shared__ int4 job_lds[2];
    if (lane < 2) job_lds[lane] = int4{0, 0, 0, 8};
    bool c_edge = false;                            // the current job's tile is one of the bottom edge (fewer than 8 rows)
    int parity = 0;
    int c_total = 0, c_next = 0;
    bool c_valid = false, o_valid = false, no_more_jobs = false;

    bool active = false;
    unsigned int rays = 0;

    // path state
    f3 o = mk3(0, 0, 0), dh = mk3(0, 0, 1), thr = mk3(1, 1, 1);
    float len = 1.0f;
    float self_t = 0.0f;
    int bounce = 0, prev = -1;
    int sb = 0;   // where the path's radiance goes: pixel slot of the tile (0..63) + 64 x LDS bank (0, 1), or 128: straight to the frame
    uint32_t pixel = 0, sample = 0, top_pixel = 0;   // pixel: counted from the bottom row like the reference's j; top_pixel: index into the frame

    // MODE 1 (grid of several layers) and 2 (one layer of cells) only: the walk through the grid of this lane's ray, and whether its answer waits to be used
    GridWalk gw;
    gw.tmx = gw.tmy = gw.tmz = gw.tdx = gw.tdy = gw.tdz = gw.t_out = gw.best_t = 0.0f;
    gw.best_i = -1; gw.ci = gw.sx = gw.sy = gw.sz = 0; gw.k = gw.kend = 0u; gw.prev_in_grid = false; gw.live = false;
    bool walking = false, pending = false;

    // One iteration = one ray of every live path: scan, then either the sky (path ends) or the hit
    // record; lanes whose path ended take a new (sample, pixel) item; ONE philox block per lane then
    // feeds either the camera ray of a new path (bounce 0) or the scatter of a continuing one.
    for (;;) {
        HitRec h;
        MatRec mat;
        bool hit = false;
        if constexpr (MODE != 0) {
            // Walks differ in length from lane to lane. Rather than wait for the longest, the wave walks until
            // at most a quarter of its rays are still on their way, handles the answers of the others (below),
            // gives those lanes their next rays and comes back here; the walks still in progress simply go on.
            for (;;) {
                const int nw = __popcll(__builtin_amdgcn_ballot_w64(walking)), np = __popcll(__builtin_amdgcn_ballot_w64(pending));
                if (nw * PT_GRID_RATIO <= np) break;
#ifdef PT_STATS
                { const int nh = __popcll(__builtin_amdgcn_ballot_w64(walking && gw.k < gw.kend));
                  PT_STAT_ADD(5, 1); PT_STAT_ADD(6, nw); PT_STAT_ADD(7, nh); }
#endif
                if (walking) {
                    grid_pass<MODE == 2>(P.grid, o, dh, 0.001f * len, prev, gw);
                    if (!gw.live) { walking = false; pending = true; }
                }
            }
        }
        const bool answered = MODE == 0 ? active : pending;      // lanes whose ray has (MODE 0: gets now) its closest hit
#ifdef PT_STATS
        if (__any(answered)) PT_STAT_ADD(9, 1);                  // iterations with a scan
#endif
        if (answered) {
            ++rays;
            float bt; int bi;
            if constexpr (MODE == 0) scan_spheres(P.scene, o, dh, 0.001f * len, prev, self_t, bt, bi);
            else { grid_end(gw, 0.001f * len, prev, self_t, bt, bi); pending = false; }
#ifdef PT_STATS
            { const unsigned long long mm = __ballot(bi < 0), hm = __ballot(bi >= 0);
              if (mm) PT_STAT_ADD(10, 1); if (hm) PT_STAT_ADD(11, 1);           // iterations that run the sky block / the finish block
              PT_STAT_ADD(12, __popcll(mm)); PT_STAT_ADD(13, __popcll(hm)); }     // lanes in them
#endif
            if (bi < 0) {                                                            // Path Tracer.cpp:105-107
                const f3 c = thr * sky(dh.y);
                if (sb < 128) {
                    unsigned long long* a = acc + sb * 3;                            // bank b starts at word 192 b = 3 x 64 b
                    atomicAdd(a + 0, to_fx(c.x));
                    atomicAdd(a + 1, to_fx(c.y));
                    atomicAdd(a + 2, to_fx(c.z));
                } else {
                    // a straggler whose job's sums have gone to the frame already (see the job switch below)
                    const size_t plane = size_t(P.W) * size_t(P.H);
                    unsigned long long* dst = P.accum_fx + top_pixel;
                    const unsigned long long vx = to_fx(c.x), vy = to_fx(c.y), vz = to_fx(c.z);
                    if (vx) atomicAdd(dst, vx);
                    if (vy) atomicAdd(dst + plane, vy);
                    if (vz) atomicAdd(dst + 2 * plane, vz);
                }
                active = false;
            } else {
                h = finish_hit(P.scene, bi, bi == prev, o, dh, 0.001 * double(len));
                mat = P.scene.mat[bi];
                prev = bi;
                ++bounce;
                // the next ray would enter calcColour with depth <= 0: black whatever the scatter (Path Tracer.cpp:86)
                hit = active = bounce < P.max_depth;
            }
        }

        // ---- hand new items to the lanes whose path has ended (regeneration) ----
        // Straight-line: the lanes that need an item are ranked; the first `rem` of them take what the current job has
        // left, and if that is not enough the wave fetches the next job NOW and the others draw from it in the same
        // pass. A lane left without an item (the new job is a small one of the bottom edge, or the item lies in the
        // part of a tile that sticks out on the right) simply asks again in the next iteration.
        bool fresh = false;
        int px = 0, j = 0;
        const unsigned long long need_mask = __ballot(!active);
        if (need_mask != 0ull && !(no_more_jobs && c_next >= c_total)) {               // wave-uniform
            const int n_need = __popcll(need_mask);
            const int rem = c_total - c_next;                                        // items left in the current job
            int n_tx = 0, n_ty = 0, n_s0 = 0, n_vh = 8, n_total = 0;
            bool have_new = false;
            if (rem < n_need && !no_more_jobs) {
                // The other bank may still have paths in flight (a path of 30 bounces outlives the whole next job).
                // Waiting for them would idle every finished lane; instead its sums go to the frame now and the
                // stragglers add their result straight to the frame when they end (integer sums: any order).
                if (active && (sb >> 6) != parity) sb = 128;
                if (o_valid) flush_bank(P, acc + 192 * (parity ^ 1), job_lds[parity ^ 1].x, job_lds[parity ^ 1].y, lane);
                // A wave's first job is its own number (the launch has at most as many waves as jobs); the counter hands
                // out the rest. Thousands of waves asking one address at the same moment would queue for ~0.1 ms.
                int job = int(blockIdx.x);
                if (c_valid) {
                    // Counter k hands out the jobs k, k + K, k + 2 K, ... (K counters on lines of their own: one address serves
                    // about 88 requests per microsecond, too few for one-sample jobs at the end of a small frame). A wave
                    // stays with the counter of its number until that one runs out, then moves to the next; no job is
                    // left when all K have been asked in vain.
                    if (lane == 0) {
                        unsigned k = my_counter;
                        for (int tries = 0; tries < PT_JOB_COUNTERS; ++tries) {
                            job = int(atomicAdd(job_counter(P, k), 1u) * PT_JOB_COUNTERS + k);
                            if (job < P.n_jobs) break;
                            k = (k + 1) % PT_JOB_COUNTERS;
                        }
                        my_counter = k;
                    }
                    job = __builtin_amdgcn_readfirstlane(job);
                }
                if (job >= P.n_jobs) {
                    no_more_jobs = true;
#if PT_PRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef PT_WAVEPROF
                    if (lane == 0) rp[1] = wall_clock64();
#endif
                } else {
                    have_new = true;
                    const int chunk = job / P.n_tiles, tile = job - chunk * P.n_tiles;
                    n_ty = tile / P.tiles_x; n_tx = tile - n_ty * P.tiles_x;
                    const bool fine = chunk >= P.n_coarse;
#if PT_PRIO
                    if (fine) __builtin_amdgcn_s_setprio(1);
#endif
                    n_s0 = fine ? P.fine_begin + (chunk - P.n_coarse) * P.spp_fine : P.spp_begin + chunk * P.spp_chunk;
                    const int s_end = fine ? min(n_s0 + P.spp_fine, P.spp_end) : min(n_s0 + P.spp_chunk, P.fine_begin);
                    n_vh = min(8, P.H - 8 * n_ty);                                   // rows of the tile inside the image
                    n_total = 8 * n_vh * (s_end - n_s0);
                    if (lane == 0) job_lds[parity ^ 1] = int4{n_tx, n_ty, n_s0, n_vh};   // (the draining job's sums have just gone to the frame)
                }
            }
            const int rank = __builtin_amdgcn_mbcnt_hi(uint32_t(need_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(need_mask), 0));
            const bool from_new = rank >= rem;
            const int k = from_new ? rank - rem : c_next + rank;                     // item number within its job
            // Straight-line at lane level too (selects, then ONE guarded block of assignments): nested conditions cost a
            // chain of register copies at every level for the ten values a new path sets.
            const int slot = (from_new ? 1 : 0) ^ parity;                            // job slot and bank of sums of this lane's item
            const int4 jb = job_lds[slot];                                           // tx, ty, s0, vh of that job
            const int tx = jb.x, ty = jb.y, s0 = jb.z;
            // item k = (sample, row, column) with the column fastest, over the rows of the tile that exist
            int row = (k >> 3) & 7, smp = k >> 6;
            if (c_edge || n_vh != 8) {                                               // wave-uniform: a tile of the bottom edge
                const int vh = jb.w;
                const int t = k >> 3;
                static_assert(8 * PT_MAX_SPP_CHUNK <= (1 << 20), "t = k >> 3 < vh * spp_chunk must stay below 2^20 for the reciprocal quotient");
                smp = int((float(t) + 0.5f) * __builtin_amdgcn_rcpf(float(vh)));      // t / vh for vh <= 8, t < 2^20: the quotient is at least 1/16 from an integer
                row = t - smp * vh;
            }
            const int x = tx * 8 + (k & 7), py = ty * 8 + row;                       // py from the top
            // (a bottom-edge tile's job may hold fewer than 64 items; max_depth >= 1: the host launches nothing for 0)
            if (!active & (!from_new | (have_new & (k < n_total))) & (x < P.W)) {
                sb = row * 8 + (k & 7) + 64 * slot;
                px = x;
                j = P.H - 1 - py;                                                    // Path Tracer.cpp:298
                pixel = uint32_t(j) * uint32_t(P.W) + uint32_t(px);
                sample = uint32_t(s0 + smp);
                bounce = 0;
                prev = -1;
                top_pixel = uint32_t(py) * uint32_t(P.W) + uint32_t(px);
                fresh = active = true;
            }
            if (have_new) {
                o_valid = c_valid;
                parity ^= 1;                                                         // the new job's slot and bank become the current ones
                c_edge = n_vh != 8;
                c_total = n_total;
                c_next = min(n_need - max(rem, 0), n_total);
                c_valid = true;
            } else {
                c_next += min(n_need, max(rem, 0));
            }
        }
        if constexpr (MODE != 0) {
            if (!__any(active)) {
                if (no_more_jobs) break;
                continue;
            }
        }                                                    // MODE 0: the block below is skipped, the check at the top of the loop ends the wave
        { const int live = __popcll(__ballot(MODE == 0 ? active : (active && !walking))); (void)live;
          PT_STAT_ADD(0, 1);                                 // wave-level ray segments
          PT_STAT_ADD(1, live); }                            // live lanes in them

        // ---- the next ray of every live path (MODE 1: of those whose previous ray has been answered) ----
        if (active && !walking) {
            // The ten round keys are wave-uniform. Left alone, hipcc computes them once, runs out of scalar registers
            // and parks them in vector lanes (20 v_readlane per iteration); opaque copies of the key make it
            // rebuild them here with scalar adds, which cost the vector ALU nothing.
            uint32_t key0 = P.key0, key1 = P.key1;
            asm volatile("" : "+s"(key0), "+s"(key1));
            const u4 r = philox4x32_10(pixel, sample, uint32_t(bounce), 0u, key0, key1);
            f3 d;
            if (fresh) {
                // jitter, then a polar map of two uniforms onto the unit disk (same distribution as
                // the rejection loop of VectorFunc.cpp:57-64)
                const float rad = fast_sqrt(u24(r.z)), ang = u24(r.w);
                // The camera is read here, through the scalar cache, rather than carried in scalar registers for the
                // whole kernel: the scan needs those registers, and hipcc would park the camera in vector lanes and
                // fetch it back value by value with vector instructions. The opaque copy of the pointer keeps the
                // loads inside this block.
                const float* cam_ptr = P.cam_dev;
                asm volatile("" : "+s"(cam_ptr));
                const CameraDev cam = load_camera(cam_ptr);
                gen_ray(cam, px, j, P.inv_w1, P.inv_h1, u24(r.x), u24(r.y), rad * cos_rev(ang), rad * sin_rev(ang), o, d);
                thr = mk3(1, 1, 1);
                if constexpr (MODE != 0) dh = normalise(d, len);
            }
            bool ok = true;
            if (hit) {
                f3 atten;
                ok = shade(mat, dh, len, h.n, h.front, u24(r.x), u24(r.y), u24(r.z), d, atten);
                thr = thr * atten;                                                   // Path Tracer.cpp:97
                o = h.p;
                if constexpr (MODE != 0) dh = normalise(d, len);
            }
            // MODE 0: one normalise for both kinds of new ray (eight vector instructions fewer per iteration; the grid
            // kernels have no register to spare for the longer life of `d` and keep one in each branch)
            if constexpr (MODE == 0) { if (fresh | hit) dh = normalise(d, len); }
            if (hit) {
                self_t = self_far_root(h.r, h.n, h.front, dh);
                active = ok;                                                         // Path Tracer.cpp:100 -> black
            }
            if constexpr (MODE != 0) {
                if (active) {
                    grid_begin(P.grid, o, dh, 0.001f * len, prev, self_t, gw);
                    walking = gw.live;
                    pending = !gw.live;                                              // the big list alone answers this ray
                }
            }
        }
        if constexpr (MODE == 0) {
            // End of the frame: no job is left and this wave holds only a few paths (the long ones). They go to k_tail,
            // which finishes them with the spheres spread over the lanes. (Tested at the bottom of the loop: at the top
            // hipcc kept two copies of the path state and copied one to the other and back in every iteration.)
            if (no_more_jobs && __popcll(__ballot(active)) <= PT_TAIL_PATHS) break;
        }
    }

    // ---- the last few paths of this wave go to its slots for k_tail (tail of the frame); the other slots are marked empty ----
    // (Finishing them here instead, with finish_paths on the wave's own sixteen, was measured in round 3: the same frame
    // time. What the end of a small frame waits for is the longest path, about 3 us per bounce either way, not the kernel boundary.)
    if constexpr (MODE == 0) {
        const unsigned long long live = __ballot(active);
        const int n_live = __popcll(live);
        const int rank = __builtin_amdgcn_mbcnt_hi(uint32_t(live >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(live), 0));
        TailPath* slots = tail_slots(P) + size_t(wave_number) * PT_TAIL_PATHS;
        if (active) {
            TailPath tp;
            tp.o[0] = o.x; tp.o[1] = o.y; tp.o[2] = o.z;
            tp.dh[0] = dh.x; tp.dh[1] = dh.y; tp.dh[2] = dh.z;
            tp.thr[0] = thr.x; tp.thr[1] = thr.y; tp.thr[2] = thr.z;
            tp.len = len; tp.self_t = self_t; tp.prev = prev; tp.bounce = bounce;
            tp.pixel = pixel; tp.sample = sample; tp.top_pixel = top_pixel;
            slots[rank] = tp;
        }
        if (lane >= n_live && lane < PT_TAIL_PATHS) slots[lane].bounce = -1;
    }

    // ---- the last sums to the frame ----
    if (o_valid) flush_bank(P, acc + 192 * (parity ^ 1), job_lds[parity ^ 1].x, job_lds[parity ^ 1].y, lane);
    if (c_valid) flush_bank(P, acc + 192 * parity, job_lds[parity].x, job_lds[parity].y, lane);
    unsigned int r = rays;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) r += __shfl_xor(r, s);
#ifdef PT_WAVEPROF
    if (lane == 0 && blockIdx.x < 8192) {
        g_rprof[4 * blockIdx.x] = rp[0]; g_rprof[4 * blockIdx.x + 1] = rp[1]; g_rprof[4 * blockIdx.x + 2] = wall_clock64(); g_rprof[4 * blockIdx.x + 3] = r;
        if (blockIdx.x == 0) g_rprof[8192 * 4] = gridDim.x;
    }
#endif
    if (lane == 0 && r) atomicAdd(ray_counter(P), (unsigned long long)r);
}

// The tail of a frame. When no job is left, a wave of k_render that holds at most PT_TAIL_PATHS paths (the long ones:
// dozens of bounces between a sphere and the ground, or inside glass) hands them over and exits: finishing them there
// would run the whole sequential scan for a few lanes at the latency of a lone wave (about 10 us per bounce), with most
// of the chip idle. Here a path has a GROUP of 16 lanes (four paths per wave): the lanes of a group hold the same path
// and share out the spheres (coop_scan). Same arithmetic per sphere, same hit rule, same philox blocks, integer sums:
// the frame is bit for bit what k_render alone would produce. A workgroup owns a run of consecutive slots and its
// sixteen groups take them one by one through a counter in LDS: a path of fifty bounces then holds up one group while
// the others work through the rest of the run (with a fixed slot sequence per group the longest sequence, two or three
// long paths in a row, set the length of the kernel). No global counter, so nothing for thousands of groups to queue at.
#ifndef PT_TAIL_LDS_MAX
#define PT_TAIL_LDS_MAX 65536        // a workgroup's LDS limit: the scene copies of k_tail<true> must fit
#endif
// LDS: the scan records, the double-precision spheres and the materials are first copied into LDS (scenes up to a
// few hundred spheres: 97 bytes per sphere), so that a bounce makes no trip to L2: the tail is pure latency.
#ifdef PT_WAVEPROF
__device__ unsigned long long g_tprof[8192 * 4 + 8];
#endif
template <bool LDS>
__global__ void __launch_bounds__(256, LDS ? 6 : 5)   // six waves per SIMD with the scene in LDS: the launch is sized for them (render_core)
k_tail(const RenderParams P) {
    constexpr int G = PT_TAIL_GROUP;
#ifdef PT_WAVEPROF
    const unsigned long long tp_start = wall_clock64();
    unsigned long long tp_staged = 0;
#endif
    extern __shared__ __align__(16) unsigned char tail_lds[];
    const int tid = threadIdx.x;
    const int n_rec_floats = (P.scene.n_pad + 8) * 4;
    const int nq_alloc = n_rec_floats / 16;                                // quads of the copy (the scene's and the two of the padding)
    float4* l_scan = reinterpret_cast<float4*>(tail_lds);
    GeomRec* l_geom = reinterpret_cast<GeomRec*>(tail_lds + size_t(n_rec_floats) * 4);
    MatRec* l_mat = reinterpret_cast<MatRec*>(reinterpret_cast<unsigned char*>(l_geom) + size_t(P.scene.n) * sizeof(GeomRec));
    if constexpr (LDS) {
        // the scan records transposed (QuadsLds): 16-byte word c of quad q to [c * nq_alloc + q]
        const float4* gs = reinterpret_cast<const float4*>(P.scene.scan);
        for (int i = tid; i < n_rec_floats / 4; i += 256) l_scan[(i & 3) * nq_alloc + (i >> 2)] = gs[i];
        const double2* gg = reinterpret_cast<const double2*>(P.scene.geom);
        for (int i = tid; i < P.scene.n * 3; i += 256) reinterpret_cast<double2*>(l_geom)[i] = gg[i];
        const float4* gm = reinterpret_cast<const float4*>(P.scene.mat);
        for (int i = tid; i < P.scene.n * 2; i += 256) reinterpret_cast<float4*>(l_mat)[i] = gm[i];
        __syncthreads();
    }
    const QuadsLds quads_lds{l_scan, nq_alloc};
    const QuadsGlobal quads_global{P.scene.scan};
#ifdef PT_WAVEPROF
    tp_staged = wall_clock64();
#endif
    __shared__ int run_next;                                              // next slot of this workgroup's run, relative to its start
    if (tid == 0) run_next = 0;
    __syncthreads();
    const int per_block = (P.n_tail_slots + int(gridDim.x) - 1) / int(gridDim.x);
    const int run_begin = int(blockIdx.x) * per_block, run_end = min(run_begin + per_block, P.n_tail_slots);
    unsigned int rays;
#ifdef PT_WAVEPROF
    unsigned long long tp_iters = 0;
#endif
#ifdef PT_WAVEPROF
    unsigned long long* const it = &tp_iters;
#else
    unsigned long long* const it = nullptr;
#endif
    if constexpr (LDS) rays = finish_paths<G>(P, tail_slots(P) + run_begin, run_end - run_begin, &run_next, quads_lds, l_geom, l_mat, tid, it);
    else rays = finish_paths<G>(P, tail_slots(P) + run_begin, run_end - run_begin, &run_next, quads_global, P.scene.geom, P.scene.mat, tid, it);
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) rays += __shfl_xor(rays, s);
#ifdef PT_WAVEPROF
    const unsigned long long tp_loop = wall_clock64();
#endif
    if ((tid & 63) == 0 && rays) atomicAdd(ray_counter(P), (unsigned long long)rays);
#ifdef PT_WAVEPROF
    if ((threadIdx.x & 63) == 0) {
        const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (w < 8192) { g_tprof[4 * w] = tp_start; g_tprof[4 * w + 1] = tp_staged; g_tprof[4 * w + 2] = tp_loop; g_tprof[4 * w + 3] = tp_iters; }
        if (w == 0) g_tprof[8192 * 4] = gridDim.x * 4;
    }
#endif
}

__global__ void k_resolve(unsigned long long* fx, float* out, size_t n_pixels, const unsigned long long* call_rays,
                          unsigned long long* total_rays) {
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_pixels; i += stride) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned long long v = fx[c * n_pixels + i];
            if (v) {
                out[3 * i + c] += float(double(v) * (1.0 / 4294967296.0));
                fx[c * n_pixels + i] = 0;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *total_rays += *call_rays;
}

// writeColour (Path Tracer.cpp:62-74) per channel, in double like the reference.
__global__ void k_quantise(const float* sum, size_t n, double scale, unsigned char* out) {
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
        double v = __builtin_sqrt(scale * double(sum[i]));         // :69-71
        v = v < 0.0 ? 0.0 : (v > 0.999 ? 0.999 : v);               // std::clamp(v, 0.0, 0.999), :73
        out[i] = (unsigned char)(int)(256.0 * v);
    }
}

template <int MODE>
__global__ void k_scan(SceneDev sc, GridDev grid, int n, const double* o, const double* d, double t_min, double* t, int* idx,
                       double* p, double* nrm, int* front) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const f3 ro = mk3(float(o[3 * k]), float(o[3 * k + 1]), float(o[3 * k + 2]));
    const f3 rd = mk3(float(d[3 * k]), float(d[3 * k + 1]), float(d[3 * k + 2]));
    float len;
    const f3 dh = normalise(rd, len);
    float bt; int bi;
    if constexpr (MODE == 0) scan_spheres(sc, ro, dh, float(t_min) * len, -1, 0.0f, bt, bi);
    else scan_grid<MODE == 2>(grid, ro, dh, float(t_min) * len, -1, 0.0f, bt, bi);
    idx[k] = bi;
    if (bi < 0) {
        t[k] = 0; front[k] = 0;
        for (int c = 0; c < 3; ++c) { p[3 * k + c] = 0; nrm[3 * k + c] = 0; }
        return;
    }
    const HitRec h = finish_hit(sc, bi, false, ro, dh, t_min * double(len));
    t[k] = h.t_unit / double(len);
    p[3 * k] = h.px; p[3 * k + 1] = h.py; p[3 * k + 2] = h.pz;
    nrm[3 * k] = h.nx; nrm[3 * k + 1] = h.ny; nrm[3 * k + 2] = h.nz;
    front[k] = h.front ? 1 : 0;
}

// calcColour (Path Tracer.cpp:82-108) for one given ray per thread with the uniform numbers of every scatter injected:
// the loop body of k_render (scan -> sky | finish, depth rule, shade, next ray) without philox, regeneration and sums.
// u is [n][max_depth][3]: the scatter at the k-th hit of a path takes u[k] = (u1, u2: sphere point; u3: Fresnel coin).
template <int MODE>
__global__ void k_trace(SceneDev sc, GridDev grid, int n, const double* o, const double* d, int max_depth, const double* u,
                        double* rgb, int* n_rays) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    f3 ro = mk3(float(o[3 * k]), float(o[3 * k + 1]), float(o[3 * k + 2]));
    float len;
    f3 dh = normalise(mk3(float(d[3 * k]), float(d[3 * k + 1]), float(d[3 * k + 2])), len);
    f3 thr = mk3(1, 1, 1), c = mk3(0, 0, 0);
    float self_t = 0.0f;
    int prev = -1, bounce = 0, rays = 0;
    bool active = max_depth > 0;                                              // Path Tracer.cpp:86
    while (active) {
        ++rays;
        float bt; int bi;
        if constexpr (MODE == 0) scan_spheres(sc, ro, dh, 0.001f * len, prev, self_t, bt, bi);
        else scan_grid<MODE == 2>(grid, ro, dh, 0.001f * len, prev, self_t, bt, bi);
        if (bi < 0) { c = thr * sky(dh.y); break; }                           // :105-107
        const HitRec h = finish_hit(sc, bi, bi == prev, ro, dh, 0.001 * double(len));
        const MatRec mat = sc.mat[bi];
        prev = bi;
        ++bounce;
        if (bounce >= max_depth) break;                                       // the next ray would enter with depth <= 0 (:86): black
        const double* uu = u + (size_t(k) * size_t(max_depth) + size_t(bounce - 1)) * 3;
        f3 dd, atten;
        const bool ok = shade(mat, dh, len, h.n, h.front, float(uu[0]), float(uu[1]), float(uu[2]), dd, atten);
        thr = thr * atten;                                                    // :97
        ro = h.p;
        dh = normalise(dd, len);
        self_t = self_far_root(h.r, h.n, h.front, dh);
        active = ok;                                                          // :100 -> black
    }
    rgb[3 * k] = c.x; rgb[3 * k + 1] = c.y; rgb[3 * k + 2] = c.z;
    n_rays[k] = rays;
}

__global__ void k_scatter(int n, const int* kind, const double* albedo, const double* fuzz, const double* ior,
                          const double* d_in, const double* nrm, const int* front, const double* u, double* d_out,
                          double* atten, int* ok) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    MatRec m;
    m.ar = float(albedo[3 * k]); m.ag = float(albedo[3 * k + 1]); m.ab = float(albedo[3 * k + 2]);
    if (kind[k] == 2) m.ar = m.ag = m.ab = 1.0f;                   // as pt_set_scene stores a Dielectric (Dielectric.cpp:7)
    m.fuzz = float(fuzz[k]);
    const double i = ior[k];
    m.ior = float(i);
    m.inv_ior = i != 0.0 ? float(1.0 / i) : 0.0f;
    const double r0 = (1.0 - i) / (1.0 + i);
    m.r0 = float(r0 * r0);
    m.kind = kind[k];
    float len;
    const f3 dh = normalise(mk3(float(d_in[3 * k]), float(d_in[3 * k + 1]), float(d_in[3 * k + 2])), len);
    const f3 nn = mk3(float(nrm[3 * k]), float(nrm[3 * k + 1]), float(nrm[3 * k + 2]));
    f3 out, a;
    const bool r = shade(m, dh, len, nn, front[k] != 0, float(u[3 * k]), float(u[3 * k + 1]), float(u[3 * k + 2]), out, a);
    d_out[3 * k] = out.x; d_out[3 * k + 1] = out.y; d_out[3 * k + 2] = out.z;
    atten[3 * k] = a.x; atten[3 * k + 1] = a.y; atten[3 * k + 2] = a.z;
    ok[k] = r ? 1 : 0;
}

__global__ void k_gen_ray(CameraDev cam, int n, float inv_w1, float inv_h1, const int* pi, const int* pj,
                          const double* u1, const double* u2, const double* diskx, const double* disky, double* o,
                          double* d) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    f3 ro, rd;
    gen_ray(cam, pi[k], pj[k], inv_w1, inv_h1, float(u1[k]), float(u2[k]), float(diskx[k]), float(disky[k]), ro, rd);
    o[3 * k] = ro.x; o[3 * k + 1] = ro.y; o[3 * k + 2] = ro.z;
    d[3 * k] = rd.x; d[3 * k + 1] = rd.y; d[3 * k + 2] = rd.z;
}

__global__ void k_sky(int n, const double* d, double* rgb) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    float len;
    const f3 dh = normalise(mk3(float(d[3 * k]), float(d[3 * k + 1]), float(d[3 * k + 2])), len);
    const f3 c = sky(dh.y);
    rgb[3 * k] = c.x; rgb[3 * k + 1] = c.y; rgb[3 * k + 2] = c.z;
}

__global__ void k_philox(int n, const uint32_t* ctr, uint32_t k0, uint32_t k1, uint32_t* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const u4 r = philox4x32_10(ctr[4 * k], ctr[4 * k + 1], ctr[4 * k + 2], ctr[4 * k + 3], k0, k1);
    out[4 * k] = r.x; out[4 * k + 1] = r.y; out[4 * k + 2] = r.z; out[4 * k + 3] = r.w;
}

// ---- pt_render_noalloc: the job sums and counters of the sample loop in a fixed array ----
// Device memory the sample loop needs beyond its inputs and outputs: the planar u64[3][H][W] job sums and two
// counters (rays of the call, next job). pt_render keeps them in buffers it allocates; pt_render_noalloc uses this
// array instead, one copy per GPU, held until the process ends (PT_NOALLOC_MAX_PIXELS from include/pt_hip.h:
// 8.4 MB + 49.8 MB). The counters and the slots between k_render<0> and k_tail (RenderParams::counters), then the sums.
#define PT_SCRATCH_COUNTER_WORDS PT_COUNTER_WORDS
#define PT_MAX_DEVICES 64
#define PT_SCRATCH_TAIL_ENTRIES (8192 * PT_TAIL_PATHS)          // 8 waves on each of the 1,024 SIMDs, PT_TAIL_PATHS slots each: 8.4 MB
#define PT_SCRATCH_HEAD_WORDS (PT_SCRATCH_COUNTER_WORDS + PT_SCRATCH_TAIL_ENTRIES * (sizeof(TailPath) / 8))
static __device__ __align__(256) unsigned long long scratch[PT_SCRATCH_HEAD_WORDS + 3 * size_t(PT_NOALLOC_MAX_PIXELS)];

// =============================================================================================
// Host side: the context and the C-ABI
// =============================================================================================
struct pt_context {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    // scene
    float* d_scan = nullptr;
    GeomRec* d_geom = nullptr;
    MatRec* d_mat = nullptr;
    int n = 0, n_pad = 0;
    bool have_scene = false, have_camera = false;
    // optional grid search (pt_set_accel): built from the host copy of the geometry on first use
    int accel = PT_ACCEL_NONE;
    bool grid_ready = false;
    std::vector<double> h_cx, h_cy, h_cz, h_r;
    uint32_t* d_cell_start = nullptr;
    float4* d_items = nullptr;
    int* d_item_ids = nullptr;
    int* d_big_ids = nullptr;
    int* d_big_local = nullptr;
    float* d_big_scan = nullptr;
    float* d_grid_shape = nullptr;
    GridDev grid;
    double grid_cell = 0;
    int grid_info[6] = {0, 0, 0, 0, 0, 0};     // nx, ny, nz, spheres in the big list, entries, fullest cell
    double grid_cell_scale = 4.0;              // cell edge = scale x median radius (PT_GRID_CELL_SCALE)
    double grid_big_ratio = 3.0;               // radius above ratio x median: big list (PT_GRID_BIG_RATIO)
    int resident_waves_grid = 0;
    bool grid_flat_ok = true;                  // PT_GRID_FLAT=0: always the general walk (diagnostic)
    CameraDev cam;
    float* d_cam = nullptr;                     // device copy of `cam` (19 floats), read by k_render
    // frame
    unsigned long long* d_fx = nullptr;
    size_t fx_elems = 0;
    float* d_accum = nullptr;
    int acc_w = 0, acc_h = 0;
    unsigned char* d_ut = nullptr;              // arena of the unit-test entry points (pt_scan, pt_scatter, ...): kept, grows only
    size_t ut_cap = 0, ut_used = 0;
    unsigned char* d_rgb8 = nullptr;            // output stage (pt_quantise_rgb8): kept between frames, grows only
    size_t rgb8_cap = 0;
    unsigned long long* d_counters = nullptr;   // RenderParams::counters: PT_COUNTER_WORDS words ([3] = rays since creation) and the slots of k_tail
    size_t tail_cap = 0;                        // capacity of that queue in entries
    int resident_waves = 0;                     // waves the device holds at once (persistent grid size)
    int n_cus = 256;
    // kernel timing
    std::vector<hipEvent_t> ev;                 // pairs
    size_t ev_used = 0;
    int target_jobs = 262144;
    int min_chunk = 3;                          // PT_MIN_CHUNK: fewest samples per pixel in a job
    int tail_rounds = 1;                        // PT_TAIL_ROUNDS: workgroups of k_tail, in units of what the chip holds at a time
    int fine_chunk = 2;                         // PT_FINE_CHUNK: samples per pixel in the small jobs at the end of a launch (0: none)
    int fine_share_pct = 100;                   // PT_FINE_SHARE: their share, in percent of one large job per wave
    int grid_factor_pct = 100;                  // PT_GRID_PCT: persistent grid as a percentage of the resident waves
};

static thread_local std::string g_create_err;

// Address of `scratch` on each device (every GPU holds its own copy at its own address), taken once per device in
// pt_create with that device current. Host code never uses the array's name for anything else: the name is a
// host-side stand-in, not a device address.
static std::mutex g_scratch_mutex;
static unsigned long long* g_scratch_ptr[PT_MAX_DEVICES];

static int fail(pt_handle h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}
#define PT_HIP(h, call)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(h, PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));          \
    } while (0)

template <class T>
static int upload(pt_handle h, T** dst, const std::vector<T>& src) {
    if (*dst) { PT_HIP(h, hipFree(*dst)); *dst = nullptr; }
    PT_HIP(h, hipMalloc(reinterpret_cast<void**>(dst), sizeof(T) * src.size()));
    PT_HIP(h, hipMemcpyAsync(*dst, src.data(), sizeof(T) * src.size(), hipMemcpyHostToDevice, h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

// Scoped device buffer for the unit-test entry points.
// A device buffer of one unit-test call: a piece of the handle's arena (pt_context::d_ut), which is kept between calls
// and only grows, so that the host classes, which ask the device one ray or one scatter at a time
// (host/Material.h, host/HittableList.h), do not pay seven to eleven hipMalloc / hipFree per call.
struct DevBuf {
    void* p = nullptr;
    template <class T> T* as() { return static_cast<T*>(p); }
};

static SceneDev scene_dev(pt_handle h) {
    SceneDev s;
    s.scan = h->d_scan; s.geom = h->d_geom; s.mat = h->d_mat; s.n = h->n; s.n_pad = h->n_pad;
    return s;
}

// Scan records (see SceneDev) of the spheres `which` lists, padded to a multiple of four plus the two
// never-hit quads the pipelined loads may touch.
static std::vector<float> scan_records(const std::vector<int>& which, const double* cx, const double* cy, const double* cz,
                                       const double* r, int* n_pad_out) {
    const int n = int(which.size());
    const int n_pad = (n + 3) & ~3;
    std::vector<float> scan(size_t(n_pad + 8) * 4);
    for (int i = 0; i < n_pad + 8; ++i) {
        float* rec = scan.data() + 8 * (i >> 1) + (i & 1);
        if (i < n) {
            const int s = which[i];
            rec[0] = float(cx[s]); rec[2] = float(cy[s]); rec[4] = float(cz[s]);
            // -K from the float centre the scan multiplies with, so that b*b + 2 o.c - |o|^2 - K is
            // the discriminant of the sphere tier 1 actually sees; r^2 in double.
            const double fx = rec[0], fy = rec[2], fz = rec[4];
            rec[6] = float(r[s] * r[s] - (fx * fx + fy * fy + fz * fz));   // -K
        } else {
            rec[0] = rec[2] = rec[4] = 0.0f;
            rec[6] = -1e30f;                     // -K: never flagged
        }
    }
    *n_pad_out = n_pad;
    return scan;
}

// The uniform grid of pt_grid.h from the host copy of the geometry. Cell edge = grid_cell_scale x the
// median radius; a sphere much larger than the median (grid_big_ratio), or whose box would span more than
// four cells on an axis, goes to the big list: it would stretch the box of the grid and fill many cells.
static int build_grid(pt_handle h) {
    const int n = h->n;
    const double *cx = h->h_cx.data(), *cy = h->h_cy.data(), *cz = h->h_cz.data(), *r = h->h_r.data();
    std::vector<double> rs(h->h_r);
    std::nth_element(rs.begin(), rs.begin() + n / 2, rs.end());
    double cell = h->grid_cell_scale * rs[n / 2];
    std::vector<int> big, small;
    double lo[3], hi[3];
    int dims[3];
    for (int attempt = 0;; ++attempt) {
        big.clear(); small.clear();
        for (int i = 0; i < n; ++i) ((r[i] > h->grid_big_ratio * rs[n / 2] || 2.0 * r[i] > 3.0 * cell) ? big : small).push_back(i);
        for (int a = 0; a < 3; ++a) { lo[a] = 0; hi[a] = 0; }
        bool first = true;
        for (int i : small) {
            const double c[3] = {cx[i], cy[i], cz[i]};
            for (int a = 0; a < 3; ++a) {
                if (first || c[a] - r[i] < lo[a]) lo[a] = c[a] - r[i];
                if (first || c[a] + r[i] > hi[a]) hi[a] = c[a] + r[i];
            }
            first = false;
        }
        double cells = 1;
        for (int a = 0; a < 3; ++a) {
            dims[a] = std::max(1, int(std::ceil((hi[a] - lo[a]) / cell + 1e-6)));
            cells *= dims[a];
        }
        if ((dims[0] <= 1024 && dims[1] <= 1024 && dims[2] <= 1024 && cells <= double(1 << 23)) || attempt >= 40) break;
        cell *= 1.5;      // a few far-flung small spheres: coarser cells rather than an enormous table
    }
    // pad: the walk is fp32 (entry point, wall distances); boxes grow by far more than its rounding
    double maxabs = 1.0;
    for (int a = 0; a < 3; ++a) maxabs = std::max(maxabs, std::max(std::fabs(lo[a]), std::fabs(hi[a])));
    const double pad = 1e-3 * cell + 1e-5 * maxabs;
    const float gmin[3] = {float(lo[0] - pad), float(lo[1] - pad), float(lo[2] - pad)};
    const float fcell = float(cell);
    for (int a = 0; a < 3; ++a) dims[a] = std::max(1, int(std::ceil((hi[a] + pad - double(gmin[a])) / double(fcell))));
    const size_t n_cells = size_t(dims[0]) * dims[1] * dims[2];
    auto cell_range = [&](int i, int a, int& c0, int& c1) {
        const double c[3] = {cx[i], cy[i], cz[i]};
        c0 = int(std::floor((c[a] - r[i] - pad - double(gmin[a])) / double(fcell)));
        c1 = int(std::floor((c[a] + r[i] + pad - double(gmin[a])) / double(fcell)));
        c0 = std::min(std::max(c0, 0), dims[a] - 1);
        c1 = std::min(std::max(c1, 0), dims[a] - 1);
    };
    std::vector<uint32_t> start(n_cells + 1, 0u), fill;
    std::vector<float4> items;
    std::vector<int> ids;
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0 counts the entries of each cell, pass 1 fills them (list order within a cell)
        if (pass == 1) {
            uint32_t run = 0;
            for (size_t c = 0; c <= n_cells; ++c) { const uint32_t v = start[c]; start[c] = run; run += v; }
            fill.assign(start.begin(), start.end() - 1);
            items.assign(start[n_cells], float4{0, 0, 0, 0});
            ids.assign(start[n_cells], 0);
        }
        for (int i : small) {
            int x0, x1, y0, y1, z0, z1;
            cell_range(i, 0, x0, x1); cell_range(i, 1, y0, y1); cell_range(i, 2, z0, z1);
            for (int z = z0; z <= z1; ++z)
                for (int y = y0; y <= y1; ++y)
                    for (int x = x0; x <= x1; ++x) {
                        const size_t c = (size_t(z) * dims[1] + y) * dims[0] + x;
                        if (pass == 0) { ++start[c]; continue; }
                        const uint32_t at = fill[c]++;
                        items[at] = float4{float(cx[i]), float(cy[i]), float(cz[i]), float(r[i] * r[i])};
                        ids[at] = i;
                    }
        }
    }
    int fullest = 0;
    for (size_t c = 0; c < n_cells; ++c) fullest = std::max(fullest, int(start[c + 1] - start[c]));
    const int n_entries = int(start[n_cells]);
    std::vector<int> big_local(n, -1);
    for (size_t b = 0; b < big.size(); ++b) big_local[big[b]] = int(b);
    int big_pad = 0;
    std::vector<float> big_scan = scan_records(big, cx, cy, cz, r, &big_pad);
    items.push_back(float4{0, 0, 0, -1.0f}); ids.push_back(0);   // the walk reads one entry past a cell's last (and drops it)
    std::vector<int> big_ids(big);
    if (big_ids.empty()) big_ids.push_back(0);
    int rc;
    if ((rc = upload(h, &h->d_cell_start, start))) return rc;
    if ((rc = upload(h, &h->d_items, items))) return rc;
    if ((rc = upload(h, &h->d_item_ids, ids))) return rc;
    if ((rc = upload(h, &h->d_big_ids, big_ids))) return rc;
    if ((rc = upload(h, &h->d_big_local, big_local))) return rc;
    if ((rc = upload(h, &h->d_big_scan, big_scan))) return rc;
    GridDev& g = h->grid;
    g.cell_start = h->d_cell_start; g.items = h->d_items; g.ids = h->d_item_ids;
    g.big_ids = h->d_big_ids; g.big_local = h->d_big_local;
    g.big.scan = h->d_big_scan; g.big.geom = nullptr; g.big.mat = nullptr; g.big.n = int(big.size()); g.big.n_pad = big_pad;
    GridShape shape;
    for (int a = 0; a < 3; ++a) { shape.gmin[a] = gmin[a]; shape.gmax[a] = gmin[a] + fcell * float(dims[a]); }
    shape.cell = fcell; shape.inv_cell = 1.0f / fcell;
    shape.nx = dims[0]; shape.ny = dims[1]; shape.nz = dims[2];
    static_assert(sizeof(GridShape) == 11 * sizeof(float), "GridShape is 11 dwords");
    std::vector<float> shape_words(16, 0.0f);
    std::memcpy(shape_words.data(), &shape, sizeof(shape));
    if ((rc = upload(h, &h->d_grid_shape, shape_words))) return rc;
    g.shape = h->d_grid_shape;
    g.last_cell = int(n_cells) - 1;
    h->grid_cell = double(fcell);
    h->grid_info[0] = dims[0]; h->grid_info[1] = dims[1]; h->grid_info[2] = dims[2];
    h->grid_info[3] = int(big.size()); h->grid_info[4] = n_entries; h->grid_info[5] = fullest;
    h->grid_ready = true;
    return 0;
}

static void rccl_handle_created();
static void rccl_handle_destroyed();

extern "C" {

const char* pt_last_error(pt_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int pt_create(int device_index, pt_handle* out) {
    if (!out) return fail(nullptr, PT_ERR_INVALID, "pt_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, PT_ERR_HIP, std::string("pt_create: no HIP device (") + hipGetErrorString(e) +
                                             "); this library has no CPU path");
    if (device_index < 0 || device_index >= count) return fail(nullptr, PT_ERR_INVALID, "pt_create: device index out of range");
    hipDeviceProp_t prop;
    PT_HIP(nullptr, hipGetDeviceProperties(&prop, device_index));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PT_ERR_INVALID, std::string("pt_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    pt_context* h = new (std::nothrow) pt_context;
    if (!h) return fail(nullptr, PT_ERR_INVALID, "pt_create: out of host memory");
    h->device = device_index;
    e = hipSetDevice(device_index);
    if (e != hipSuccess) { delete h; return fail(nullptr, PT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e)); }
    e = hipStreamCreate(&h->own_stream);
    if (e != hipSuccess) { delete h; return fail(nullptr, PT_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    h->stream = h->own_stream;
    {
        int per_cu = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render<0>, 64, 0);
        if (e == hipSuccess) {
            h->resident_waves = std::max(1, per_cu) * prop.multiProcessorCount;
            h->n_cus = prop.multiProcessorCount;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render<1>, 64, 0);      // <2> needs no more
            h->resident_waves_grid = std::max(1, per_cu) * prop.multiProcessorCount;
        }
        if (e != hipSuccess) { (void)hipStreamDestroy(h->own_stream); delete h; return fail(nullptr, PT_ERR_HIP, std::string("hipOccupancyMaxActiveBlocksPerMultiprocessor: ") + hipGetErrorString(e)); }
    }
    if (const char* gp = std::getenv("PT_GRID_PCT")) { int v = std::atoi(gp); if (v > 0) h->grid_factor_pct = v; }
    h->tail_cap = size_t(h->resident_waves) * size_t(std::max(100, h->grid_factor_pct)) / 100 * PT_TAIL_PATHS + 64;
    const size_t counter_bytes = PT_COUNTER_WORDS * sizeof(unsigned long long) + h->tail_cap * sizeof(TailPath);
    e = hipMalloc(reinterpret_cast<void**>(&h->d_counters), counter_bytes);
    if (e == hipSuccess) e = hipMemset(h->d_counters, 0, counter_bytes);
    if (e != hipSuccess) { (void)hipStreamDestroy(h->own_stream); delete h; return fail(nullptr, PT_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    // set-up that pt_render_noalloc must not do later: the address of this device's copy of `scratch`, and the
    // number of waves the device holds at once
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (device_index < PT_MAX_DEVICES && !g_scratch_ptr[device_index]) {
            unsigned long long* sp = nullptr;
            e = hipGetSymbolAddress(reinterpret_cast<void**>(&sp), HIP_SYMBOL(scratch));
            if (e != hipSuccess || !sp) {
                (void)hipFree(h->d_counters); (void)hipStreamDestroy(h->own_stream); delete h;
                return fail(nullptr, PT_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
            }
            g_scratch_ptr[device_index] = sp;
        }
    }
    if (const char* tj = std::getenv("PT_TARGET_JOBS")) { int v = std::atoi(tj); if (v > 0) h->target_jobs = v; }
    if (const char* mc = std::getenv("PT_MIN_CHUNK")) { int v = std::atoi(mc); if (v > 0) h->min_chunk = v; }
    if (const char* tr = std::getenv("PT_TAIL_ROUNDS")) { int v = std::atoi(tr); if (v > 0) h->tail_rounds = v; }
    if (const char* fc = std::getenv("PT_FINE_CHUNK")) { int v = std::atoi(fc); if (v >= 0) h->fine_chunk = v; }
    if (const char* fs = std::getenv("PT_FINE_SHARE")) { int v = std::atoi(fs); if (v >= 0) h->fine_share_pct = v; }
    if (const char* cs = std::getenv("PT_GRID_CELL_SCALE")) { double v = std::atof(cs); if (v > 0.1 && v < 1000) h->grid_cell_scale = v; }
    if (const char* gf = std::getenv("PT_GRID_FLAT")) h->grid_flat_ok = std::atoi(gf) != 0;
    if (const char* br = std::getenv("PT_GRID_BIG_RATIO")) { double v = std::atof(br); if (v >= 1 && v < 1e6) h->grid_big_ratio = v; }
    rccl_handle_created();
    *out = h;
    return 0;
}

int pt_destroy(pt_handle h) {
    if (!h) return PT_ERR_INVALID;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    (void)hipFree(h->d_scan); (void)hipFree(h->d_geom); (void)hipFree(h->d_mat);
    (void)hipFree(h->d_fx); (void)hipFree(h->d_accum); (void)hipFree(h->d_counters); (void)hipFree(h->d_cam); (void)hipFree(h->d_rgb8); (void)hipFree(h->d_ut);
    (void)hipFree(h->d_cell_start); (void)hipFree(h->d_items); (void)hipFree(h->d_item_ids);
    (void)hipFree(h->d_big_ids); (void)hipFree(h->d_big_local); (void)hipFree(h->d_big_scan); (void)hipFree(h->d_grid_shape);
    (void)hipStreamDestroy(h->own_stream);
    delete h;
    rccl_handle_destroyed();
    return 0;
}

int pt_set_stream(pt_handle h, void* hip_stream) {
    if (!h) return PT_ERR_INVALID;
    PT_HIP(h, hipSetDevice(h->device));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return 0;
}

int pt_synchronize(pt_handle h) {
    if (!h) return PT_ERR_INVALID;
    PT_HIP(h, hipSetDevice(h->device));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pt_set_scene(pt_handle h, int n, const double* cx, const double* cy, const double* cz, const double* r,
                 const int* kind, const double* ar, const double* ag, const double* ab, const double* fuzz,
                 const double* ior) {
    if (!h) return PT_ERR_INVALID;
    if (n <= 0 || !cx || !cy || !cz || !r || !kind || !ar || !ag || !ab || !fuzz || !ior)
        return fail(h, PT_ERR_INVALID, "pt_set_scene: n must be positive and no array may be NULL");
    for (int i = 0; i < n; ++i) {
        if (kind[i] < 0 || kind[i] > 2) return fail(h, PT_ERR_INVALID, "pt_set_scene: unknown material kind");
        if (!(r[i] > 0.0) || !std::isfinite(r[i]) || !std::isfinite(cx[i]) || !std::isfinite(cy[i]) || !std::isfinite(cz[i]))
            return fail(h, PT_ERR_INVALID, "pt_set_scene: sphere with non-finite centre or non-positive radius");
        if (kind[i] == PT_DIELECTRIC && !(ior[i] > 0.0)) return fail(h, PT_ERR_INVALID, "pt_set_scene: dielectric with non-positive refractive index");
        if (kind[i] != PT_DIELECTRIC && !(ar[i] >= 0.0 && ag[i] >= 0.0 && ab[i] >= 0.0 && std::isfinite(ar[i]) && std::isfinite(ag[i]) && std::isfinite(ab[i])))
            return fail(h, PT_ERR_INVALID, "pt_set_scene: albedo must be finite and non-negative");       // the fixed-point sums rely on it
        if (kind[i] == PT_METAL && !std::isfinite(fuzz[i])) return fail(h, PT_ERR_INVALID, "pt_set_scene: non-finite fuzz");
    }
    PT_HIP(h, hipSetDevice(h->device));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    std::vector<int> all(n);
    for (int i = 0; i < n; ++i) all[i] = i;
    int n_pad = 0;
    std::vector<float> scan = scan_records(all, cx, cy, cz, r, &n_pad);
    std::vector<GeomRec> geom(n);
    std::vector<MatRec> mat(n);
    for (int i = 0; i < n; ++i) {
        geom[i] = GeomRec{cx[i], cy[i], cz[i], r[i], 1 / r[i], r[i] * r[i]};   // Sphere.cpp:9, :28
        MatRec m;
        m.ar = float(ar[i]); m.ag = float(ag[i]); m.ab = float(ab[i]);
        if (kind[i] == PT_DIELECTRIC) m.ar = m.ag = m.ab = 1.0f;                // its attenuation (Dielectric.cpp:7), whatever albedo the caller passed
        m.fuzz = float(fuzz[i] < 1 ? fuzz[i] : 1);                              // Metal.h:21
        const double ii = kind[i] == PT_DIELECTRIC ? ior[i] : 1.0;
        m.ior = float(ii); m.inv_ior = float(1.0 / ii);
        const double r0 = (1 - ii) / (1 + ii);
        m.r0 = float(r0 * r0);
        m.kind = kind[i];
        mat[i] = m;
    }
    int rc;
    if ((rc = upload(h, &h->d_scan, scan))) return rc;
    if ((rc = upload(h, &h->d_geom, geom))) return rc;
    if ((rc = upload(h, &h->d_mat, mat))) return rc;
    h->n = n; h->n_pad = n_pad; h->have_scene = true;
    h->h_cx.assign(cx, cx + n); h->h_cy.assign(cy, cy + n); h->h_cz.assign(cz, cz + n); h->h_r.assign(r, r + n);
    h->grid_ready = false;
    if (h->accel == PT_ACCEL_GRID) return build_grid(h);
    return 0;
}

int pt_set_accel(pt_handle h, int mode) {
    if (!h) return PT_ERR_INVALID;
    if (mode != PT_ACCEL_NONE && mode != PT_ACCEL_GRID) return fail(h, PT_ERR_INVALID, "pt_set_accel: unknown mode");
    PT_HIP(h, hipSetDevice(h->device));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    h->accel = mode;
    if (mode == PT_ACCEL_GRID && h->have_scene && !h->grid_ready) return build_grid(h);
    return 0;
}

int pt_accel_info(pt_handle h, int info[6], double* cell_edge) {
    if (!h || !info) return PT_ERR_INVALID;
    if (!h->grid_ready) return fail(h, PT_ERR_STATE, "pt_accel_info: no grid has been built (pt_set_accel(PT_ACCEL_GRID) and pt_set_scene first)");
    for (int i = 0; i < 6; ++i) info[i] = h->grid_info[i];
    if (cell_edge) *cell_edge = h->grid_cell;
    return 0;
}

int pt_set_camera(pt_handle h, const double position[3], const double lower_left[3], const double horizontal[3],
                  const double vertical[3], const double u[3], const double v[3], double lens_radius) {
    if (!h) return PT_ERR_INVALID;
    if (!position || !lower_left || !horizontal || !vertical || !u || !v) return fail(h, PT_ERR_INVALID, "pt_set_camera: NULL vector");
    auto f = [](const double* a) { return f3{float(a[0]), float(a[1]), float(a[2])}; };
    h->cam.pos = f(position);
    const double rel[3] = {lower_left[0] - position[0], lower_left[1] - position[1], lower_left[2] - position[2]};
    h->cam.ll_rel = f(rel);
    h->cam.hor = f(horizontal); h->cam.ver = f(vertical); h->cam.u = f(u); h->cam.v = f(v);
    h->cam.lens = float(lens_radius);
    static_assert(sizeof(CameraDev) == 19 * sizeof(float), "CameraDev is 19 packed floats");
    PT_HIP(h, hipSetDevice(h->device));
    PT_HIP(h, hipStreamSynchronize(h->stream));          // a render in flight may still read the old camera
    if (!h->d_cam) PT_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->d_cam), 32 * sizeof(float)));
    PT_HIP(h, hipMemcpy(h->d_cam, &h->cam, sizeof(CameraDev), hipMemcpyHostToDevice));
    h->have_camera = true;
    return 0;
}

int pt_clear_accum(pt_handle h, int width, int height) {
    if (!h) return PT_ERR_INVALID;
    if (width <= 0 || height <= 0 || size_t(width) * size_t(height) > (size_t(1) << 28))
        return fail(h, PT_ERR_INVALID, "pt_clear_accum: bad image size");
    PT_HIP(h, hipSetDevice(h->device));
    const size_t n = size_t(width) * height * 3;
    if (width != h->acc_w || height != h->acc_h || !h->d_accum) {
        PT_HIP(h, hipStreamSynchronize(h->stream));
        if (h->d_accum) { PT_HIP(h, hipFree(h->d_accum)); h->d_accum = nullptr; }
        PT_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->d_accum), n * sizeof(float)));
        h->acc_w = width; h->acc_h = height;
    }
    PT_HIP(h, hipMemsetAsync(h->d_accum, 0, n * sizeof(float), h->stream));
    return 0;
}

void* pt_accum_device_ptr(pt_handle h) { return h ? h->d_accum : nullptr; }

int pt_read_accum(pt_handle h, int width, int height, float* host_f32) {
    if (!h || !host_f32) return PT_ERR_INVALID;
    if (!h->d_accum) return fail(h, PT_ERR_STATE, "pt_read_accum: no accumulation buffer (call pt_clear_accum or pt_render first)");
    if (width != h->acc_w || height != h->acc_h)
        return fail(h, PT_ERR_INVALID, "pt_read_accum: the caller's array is " + std::to_string(width) + "x" + std::to_string(height) +
                                           " but the buffer is " + std::to_string(h->acc_w) + "x" + std::to_string(h->acc_h));
    PT_HIP(h, hipSetDevice(h->device));
    PT_HIP(h, hipMemcpyAsync(host_f32, h->d_accum, size_t(h->acc_w) * h->acc_h * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pt_write_accum(pt_handle h, int width, int height, const float* host_f32) {
    if (!h || !host_f32) return PT_ERR_INVALID;
    int rc = pt_clear_accum(h, width, height);
    if (rc) return rc;
    PT_HIP(h, hipMemcpyAsync(h->d_accum, host_f32, size_t(width) * height * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pt_quantise_rgb8(pt_handle h, int width, int height, int spp, const float* accum_rgb_f32, uint8_t* host_rgb8) {
    if (!h || !host_rgb8) return PT_ERR_INVALID;
    if (width <= 0 || height <= 0 || spp <= 0 || size_t(width) * size_t(height) > (size_t(1) << 28))
        return fail(h, PT_ERR_INVALID, "pt_quantise_rgb8: bad size or sample count");
    const float* src = accum_rgb_f32;
    if (!src) {
        if (!h->d_accum || h->acc_w != width || h->acc_h != height) return fail(h, PT_ERR_STATE, "pt_quantise_rgb8: the handle has no accumulation buffer of this size");
        src = h->d_accum;
    }
    PT_HIP(h, hipSetDevice(h->device));
    const size_t n = size_t(width) * height * 3;
    if (h->rgb8_cap < n) {
        PT_HIP(h, hipStreamSynchronize(h->stream));
        if (h->d_rgb8) { PT_HIP(h, hipFree(h->d_rgb8)); h->d_rgb8 = nullptr; h->rgb8_cap = 0; }
        PT_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->d_rgb8), n));
        h->rgb8_cap = n;
    }
    const unsigned blocks = unsigned(std::min<size_t>((n + 255) / 256, 256 * 16));
    hipLaunchKernelGGL(k_quantise, dim3(blocks), dim3(256), 0, h->stream, src, n, 1.0 / double(spp), h->d_rgb8);
    PT_HIP(h, hipGetLastError());
    PT_HIP(h, hipMemcpyAsync(host_rgb8, h->d_rgb8, n, hipMemcpyDeviceToHost, h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

// The sample loop of one call, shared by pt_render and pt_render_noalloc: clears the two call counters, launches
// k_render on the handle's stream (sums into fx, which must be zero), then k_resolve (fx -> out, fx back to zero).
// Allocates nothing unless `timed` needs new events. counters: RenderParams::counters (PT_COUNTER_WORDS words, then the
// slots of k_tail with room for tail_cap entries).
static int render_core(pt_handle h, int width, int height, int spp_begin, int spp_end, int max_depth, uint64_t seed, float* out,
                       unsigned long long* fx, unsigned long long* counters, size_t tail_cap, bool timed) {
    const size_t n_elems = size_t(width) * height * 3;
    PT_HIP(h, hipMemsetAsync(counters, 0, sizeof(unsigned long long), h->stream));
    const int spp = spp_end - spp_begin;
    if (spp > 0 && max_depth > 0) {          // depth 0: calcColour returns black before it traces anything (Path Tracer.cpp:86)
        RenderParams P;
        P.W = width; P.H = height;
        P.spp_begin = spp_begin; P.spp_end = spp_end;
        P.tiles_x = (width + 7) / 8;
        const long long tiles = (long long)P.tiles_x * ((height + 7) / 8);
        long long want = (h->target_jobs + tiles - 1) / tiles;
        if (want < 1) want = 1;
        if (want > spp) want = spp;
        const int min_chunk = h->min_chunk;
        if (want > (spp + min_chunk - 1) / min_chunk) want = (spp + min_chunk - 1) / min_chunk;     // at least min_chunk spp (x 64 items) per job
        P.spp_chunk = int((spp + want - 1) / want);
        // A job's item number k = (sample, row, column) is an int below 64 * spp_chunk, and a bottom-edge tile divides
        // k >> 3 by its row count with a reciprocal that is exact below 2^20 (k_render): bound the chunk (more, smaller
        // jobs instead; reached only with more than 65,536 samples per pixel and tile in one call).
        if (P.spp_chunk > PT_MAX_SPP_CHUNK) P.spp_chunk = PT_MAX_SPP_CHUNK;
        const bool use_grid = h->accel == PT_ACCEL_GRID;
        const int resident = use_grid ? h->resident_waves_grid : h->resident_waves;   // from pt_create
        // The last jobs are small ones: as many samples as one large job for every wave of the launch (a wave that took
        // the last large job is then still busy while the others work through the small ones), at most half the call.
        int fine_spp = 0;
        P.spp_fine = std::min(h->fine_chunk, P.spp_chunk);
        if (P.spp_fine > 0 && P.spp_fine < P.spp_chunk) {
            const long long waves = (long long)resident * h->grid_factor_pct / 100;
            long long f = (waves * P.spp_chunk * h->fine_share_pct / 100 + tiles - 1) / tiles;
            f = std::min<long long>((f + P.spp_fine - 1) / P.spp_fine * P.spp_fine, spp / 2 / P.spp_fine * P.spp_fine);
            fine_spp = int(std::max<long long>(f, 0));
        }
        if (fine_spp == 0) P.spp_fine = 1;
        P.fine_begin = spp_end - fine_spp;
        P.n_coarse = (spp - fine_spp + P.spp_chunk - 1) / P.spp_chunk;
        P.n_tiles = int(tiles);
        const long long jobs = tiles * (P.n_coarse + (fine_spp + P.spp_fine - 1) / P.spp_fine);
        if (jobs > 0x7fffffffLL) return fail(h, PT_ERR_INVALID, "pt_render: too many jobs for one launch");
        P.max_depth = max_depth;
        P.key0 = uint32_t(seed); P.key1 = uint32_t(seed >> 32);
        // (W - 1) and (H - 1) divide by zero for a one-pixel-wide image in the reference too
        // (Path Tracer.cpp:270-271); the float reciprocal is inf there as well.
        P.inv_w1 = 1.0f / float(width - 1); P.inv_h1 = 1.0f / float(height - 1);
        P.cam_dev = h->d_cam;
        P.scene = scene_dev(h);
        if (use_grid && !h->grid_ready) return fail(h, PT_ERR_STATE, "pt_render: the grid has not been built");
        if (use_grid) P.grid = h->grid; else std::memset(&P.grid, 0, sizeof(P.grid));
        P.accum_fx = fx;
        P.counters = counters;
        P.n_tail_slots = 0;
        P.n_jobs = int(jobs);
        const bool flat = use_grid && h->grid_info[1] == 1 && h->grid_flat_ok;       // one layer of cells: the walk without a y axis
        long long grid = std::min<long long>(jobs, (long long)resident * h->grid_factor_pct / 100);
        if (!use_grid) grid = std::min<long long>(grid, (long long)(tail_cap / PT_TAIL_PATHS));      // every wave may queue PT_TAIL_PATHS paths
        if (grid < 1) grid = 1;
        if (grid > PT_JOB_COUNTERS) grid -= grid % PT_JOB_COUNTERS;
        // a wave's first job is its own number: counter k (jobs k, k + K, ...) starts behind the jobs the waves took that way
        {
            const int each = int(grid / PT_JOB_COUNTERS), more = int(grid % PT_JOB_COUNTERS);     // `more` only for launches of fewer than K waves
            PT_HIP(h, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(counters + 8), each, 32 * PT_JOB_COUNTERS, h->stream));
            if (more) PT_HIP(h, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(counters + 8), each + 1, 32 * more, h->stream));
        }
        // three events per timed call: before k_render, between k_render<0> and k_tail, after the last kernel
        timed = timed && h->ev_used + 3 <= 3 * 4096;
        while (timed && h->ev.size() < h->ev_used + 3) {
            hipEvent_t a;
            PT_HIP(h, hipEventCreate(&a));
            h->ev.push_back(a);
        }
        if (timed) PT_HIP(h, hipEventRecord(h->ev[h->ev_used], h->stream));
        if (flat) hipLaunchKernelGGL(k_render<2>, dim3(unsigned(grid < 1 ? 1 : grid)), dim3(64), 0, h->stream, P);
        else if (use_grid) hipLaunchKernelGGL(k_render<1>, dim3(unsigned(grid < 1 ? 1 : grid)), dim3(64), 0, h->stream, P);
        else {
            hipLaunchKernelGGL(k_render<0>, dim3(unsigned(grid < 1 ? 1 : grid)), dim3(64), 0, h->stream, P);
            PT_HIP(h, hipGetLastError());
            if (timed) PT_HIP(h, hipEventRecord(h->ev[h->ev_used + 1], h->stream));
            // the long paths the waves of k_render were left with (counted into the kernel time)
            P.n_tail_slots = int((grid < 1 ? 1 : grid) * PT_TAIL_PATHS);
            const size_t lds = size_t(P.scene.n_pad + 8) * 16 + size_t(P.scene.n) * (sizeof(GeomRec) + sizeof(MatRec));
            const unsigned tail_blocks = unsigned((P.n_tail_slots + 256 / PT_TAIL_GROUP - 1) / (256 / PT_TAIL_GROUP));   // one slot per group
            // with the scene in LDS a compute unit holds 160 KB / lds workgroups at a time: no more than fill the chip once
            const unsigned lds_blocks = std::min<unsigned>(tail_blocks, unsigned(h->tail_rounds) * unsigned(h->n_cus) * unsigned(std::max<size_t>(1, (160 * 1024) / std::max<size_t>(lds, 1))));
            if (lds <= PT_TAIL_LDS_MAX) hipLaunchKernelGGL(k_tail<true>, dim3(lds_blocks), dim3(256), lds, h->stream, P);
            else hipLaunchKernelGGL(k_tail<false>, dim3(tail_blocks), dim3(256), 0, h->stream, P);
        }
        PT_HIP(h, hipGetLastError());
        if (timed) {
            if (use_grid) PT_HIP(h, hipEventRecord(h->ev[h->ev_used + 1], h->stream));     // no k_tail behind the grid kernels
            PT_HIP(h, hipEventRecord(h->ev[h->ev_used + 2], h->stream));
            h->ev_used += 3;
        }
    }
    {
        const size_t n_pixels = n_elems / 3;
        const unsigned blocks = unsigned(std::min<size_t>((n_pixels + 255) / 256, 256 * 16));
        hipLaunchKernelGGL(k_resolve, dim3(blocks), dim3(256), 0, h->stream, fx, out, n_pixels, counters, h->d_counters + 3);
        PT_HIP(h, hipGetLastError());
    }
    return 0;
}

static int check_render_args(pt_handle h, const char* who, int width, int height, int spp_begin, int spp_end, int max_depth) {
    if (!h->have_scene || !h->have_camera) return fail(h, PT_ERR_STATE, std::string(who) + ": set the scene and the camera first");
    if (width <= 0 || height <= 0 || size_t(width) * size_t(height) > (size_t(1) << 28)) return fail(h, PT_ERR_INVALID, std::string(who) + ": bad image size");
    if (spp_begin < 0 || spp_end < spp_begin) return fail(h, PT_ERR_INVALID, std::string(who) + ": bad sample range");
    if (max_depth < 0) return fail(h, PT_ERR_INVALID, std::string(who) + ": negative max_depth");
    return 0;
}

int pt_render(pt_handle h, int width, int height, int spp_begin, int spp_end, int max_depth, uint64_t seed,
              float* accum_rgb_f32, uint64_t* ray_count_out) {
    if (!h) return PT_ERR_INVALID;
    if (int rc = check_render_args(h, "pt_render", width, height, spp_begin, spp_end, max_depth)) return rc;
    PT_HIP(h, hipSetDevice(h->device));
    float* out = accum_rgb_f32;
    if (!out) {
        if (!h->d_accum || h->acc_w != width || h->acc_h != height) {
            int rc = pt_clear_accum(h, width, height);
            if (rc) return rc;
        }
        out = h->d_accum;
    }
    const size_t n_elems = size_t(width) * height * 3;
    if (h->fx_elems < n_elems) {
        PT_HIP(h, hipStreamSynchronize(h->stream));
        if (h->d_fx) { PT_HIP(h, hipFree(h->d_fx)); h->d_fx = nullptr; h->fx_elems = 0; }
        PT_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->d_fx), n_elems * sizeof(unsigned long long)));
        PT_HIP(h, hipMemsetAsync(h->d_fx, 0, n_elems * sizeof(unsigned long long), h->stream));
        h->fx_elems = n_elems;
    }
    if (int rc = render_core(h, width, height, spp_begin, spp_end, max_depth, seed, out, h->d_fx, h->d_counters, h->tail_cap, true)) return rc;
#ifdef PT_STATS
    {
        unsigned long long st[16];
        PT_HIP(h, hipStreamSynchronize(h->stream));
        PT_HIP(h, hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stats), sizeof(st)));
        if (st[9]) fprintf(stderr, "[pt stats] per scan iteration (%llu): quads %.2f, flagged quads %.2f, spheres resolved %.2f; sky block in %.3f of the iterations (%.1f lanes), finish block in %.3f (%.1f lanes)\n",
                           st[9], double(st[2]) / st[9], double(st[3]) / st[9], double(st[8]) / st[9], double(st[10]) / st[9], st[10] ? double(st[12]) / st[10] : 0.0,
                           double(st[11]) / st[9], st[11] ? double(st[13]) / st[11] : 0.0);
        fprintf(stderr, "[pt stats] wave_segments %llu live_lanes %llu (%.3f of 64) quads %llu tier2_quads %llu (%.4f) tier2_lanes %llu\n",
                st[0], st[1], st[0] ? double(st[1]) / st[0] / 64 : 0.0, st[2], st[3], st[2] ? double(st[3]) / st[2] : 0.0, st[4]);
        if (st[5]) fprintf(stderr, "[pt stats] grid passes %llu (%.2f per wave segment) walking lanes %.3f of 64, lanes testing an entry %.3f of 64\n",
                           st[5], st[0] ? double(st[5]) / st[0] : 0.0, double(st[6]) / st[5], double(st[7]) / st[5]);
        unsigned long long z[16] = {0};
        PT_HIP(h, hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z)));
    }
#endif
#ifdef PT_WAVEPROF
    {
        static std::vector<unsigned long long> st(8192 * 4 + 8);
        PT_HIP(h, hipStreamSynchronize(h->stream));
        PT_HIP(h, hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_rprof), st.size() * 8));
        const size_t nw = std::min<size_t>(8192, st[8192 * 4]);
        unsigned long long t0 = ~0ull;
        std::vector<double> nomore, ex;
        for (size_t w = 0; w < nw; ++w) t0 = std::min(t0, st[4 * w]);
        double last_start = 0, rays_sum = 0, rays_max = 0;
        std::vector<double> rw;
        for (size_t w = 0; w < nw; ++w) {
            last_start = std::max(last_start, double(st[4 * w] - t0) / 100.0);
            if (st[4 * w + 1]) nomore.push_back(double(st[4 * w + 1] - t0) / 100.0);
            ex.push_back(double(st[4 * w + 2] - t0) / 100.0);
            rays_sum += double(st[4 * w + 3]); rays_max = std::max(rays_max, double(st[4 * w + 3])); rw.push_back(double(st[4 * w + 3]));
        }
        std::sort(rw.begin(), rw.end());
        std::sort(nomore.begin(), nomore.end()); std::sort(ex.begin(), ex.end());
        auto q = [](const std::vector<double>& v, double f) { return v.empty() ? 0.0 : v[size_t(f * (v.size() - 1))]; };
        fprintf(stderr, "[render prof] waves %zu last start +%.1f us; out of jobs at (min/10%%/50%%/90%%/max) %.0f %.0f %.0f %.0f %.0f us; exit at %.0f %.0f %.0f %.0f %.0f us; rays per wave mean %.0f max %.0f, min/10%%/50%%/90%% %.0f %.0f %.0f %.0f\n",
                nw, last_start, q(nomore, 0), q(nomore, 0.1), q(nomore, 0.5), q(nomore, 0.9), q(nomore, 1), q(ex, 0), q(ex, 0.1), q(ex, 0.5), q(ex, 0.9), q(ex, 1), rays_sum / nw, rays_max, q(rw, 0), q(rw, 0.1), q(rw, 0.5), q(rw, 0.9));
    }
    {
        static std::vector<unsigned long long> st(8192 * 4 + 8);
        PT_HIP(h, hipStreamSynchronize(h->stream));
        PT_HIP(h, hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_tprof), st.size() * 8));
        const size_t nw = std::min<size_t>(8192, st[8192 * 4]);
        unsigned long long t0 = ~0ull, stage_end = 0, loop_end = 0, last_start = 0, it_max = 0, it_sum = 0, busy = 0;
        double stage_sum = 0, loop_sum = 0;
        for (size_t w = 0; w < nw; ++w) {
            t0 = std::min(t0, st[4 * w]); last_start = std::max(last_start, st[4 * w]);
            stage_end = std::max(stage_end, st[4 * w + 1]); loop_end = std::max(loop_end, st[4 * w + 2]);
            it_max = std::max(it_max, st[4 * w + 3] & 0xfffff); it_sum += st[4 * w + 3] & 0xfffff; busy += st[4 * w + 3] ? 1 : 0;
            stage_sum += double(st[4 * w + 1] - st[4 * w]); loop_sum += double(st[4 * w + 2] - st[4 * w + 1]);
        }
        {
            unsigned long long sp[16];
            PT_HIP(h, hipMemcpyFromSymbol(sp, HIP_SYMBOL(g_sprof), sizeof(sp)));
            if (sp[4]) fprintf(stderr, "[tail prof] late iterations of waves with several paths (%.2f on average): %llu, per iteration scan %.0f ns, finish %.0f ns, scatter %.0f ns\n",
                               double(sp[8]) / sp[4], sp[4], 10.0 * sp[5] / sp[4], 10.0 * sp[6] / sp[4], 10.0 * sp[7] / sp[4]);
            if (sp[0]) fprintf(stderr, "[tail prof] a path alone in its wave: %llu bounces, per bounce scan %.0f ns, finish %.0f ns, scatter %.0f ns\n", sp[0],
                               10.0 * sp[1] / sp[0], 10.0 * sp[2] / sp[0], 10.0 * sp[3] / sp[0]);
            unsigned long long z[16] = {0};
            PT_HIP(h, hipMemcpyToSymbol(HIP_SYMBOL(g_sprof), z, sizeof(z)));
        }
        {
            std::vector<double> le;
            for (size_t w = 0; w < nw; ++w) if (st[4 * w + 3]) le.push_back(double(st[4 * w + 2] - t0) / 100.0);
            std::sort(le.begin(), le.end());
            auto q = [&](double f) { return le.empty() ? 0.0 : le[size_t(f * (le.size() - 1))]; };
            std::vector<std::pair<unsigned long long, size_t>> ends;
            for (size_t w = 0; w < nw; ++w) ends.push_back({st[4 * w + 2], w});
            std::sort(ends.rbegin(), ends.rend());
            static std::vector<unsigned long long> st2(8192 * 2);
            PT_HIP(h, hipMemcpyFromSymbol(st2.data(), HIP_SYMBOL(g_tprof2), st2.size() * 8));
            for (size_t k = 0; k < std::min<size_t>(4, ends.size()); ++k) {
                const size_t w = ends[k].second;
                fprintf(stderr, "[tail prof] one of the last waves: started +%.1f us, left +%.1f us, %llu iterations, %llu of them with one path (%.1f us; the others %.1f us)\n",
                        (st[4 * w] - t0) / 100.0, (st[4 * w + 2] - t0) / 100.0, st[4 * w + 3] & 0xfffff, st[4 * w + 3] >> 20, st2[2 * w] / 100.0, st2[2 * w + 1] / 100.0);
            }
            fprintf(stderr, "[tail prof] waves leave at (10/50/90/99/99.9 %%/max) %.0f %.0f %.0f %.0f %.0f %.0f us\n", q(0.1), q(0.5), q(0.9), q(0.99), q(0.999), q(1.0));
        }
        fprintf(stderr, "[tail prof] waves %zu (with a path %llu): last start +%.1f us, last staged +%.1f us (mean staging %.1f), last loop end +%.1f us (mean loop %.1f us), iterations max %llu mean %.2f\n",
                nw, busy, (last_start - t0) / 100.0, (stage_end - t0) / 100.0, stage_sum / nw / 100.0, (loop_end - t0) / 100.0, loop_sum / nw / 100.0, it_max, double(it_sum) / nw);
    }
#endif
    if (ray_count_out) {
        unsigned long long v = 0;
        PT_HIP(h, hipMemcpyAsync(&v, h->d_counters, sizeof(v), hipMemcpyDeviceToHost, h->stream));
        PT_HIP(h, hipStreamSynchronize(h->stream));
        *ray_count_out = v;
    }
    return 0;
}

int pt_render_noalloc(pt_handle h, int width, int height, int spp_begin, int spp_end, int max_depth, uint64_t seed,
                      float* accum_rgb_f32, uint64_t* ray_count_dev) {
    if (!h) return PT_ERR_INVALID;
    if (int rc = check_render_args(h, "pt_render_noalloc", width, height, spp_begin, spp_end, max_depth)) return rc;
    if (!accum_rgb_f32) return fail(h, PT_ERR_INVALID, "pt_render_noalloc: the caller passes the accumulation buffer (a device pointer)");
    if (size_t(width) * size_t(height) > size_t(PT_NOALLOC_MAX_PIXELS))
        return fail(h, PT_ERR_INVALID, "pt_render_noalloc: image of " + std::to_string(width) + "x" + std::to_string(height) +
                                           " exceeds the fixed job-sum array (" + std::to_string(PT_NOALLOC_MAX_PIXELS) +
                                           " pixels, 1920x1080); use pt_render");
    unsigned long long* sp = h->device < PT_MAX_DEVICES ? g_scratch_ptr[h->device] : nullptr;     // set by pt_create on this device
    if (!sp) return fail(h, PT_ERR_STATE, "pt_render_noalloc: no scratch address for this device");
    PT_HIP(h, hipSetDevice(h->device));
    // The array keeps whatever the last call on this GPU left (k_resolve leaves the sums it read at zero, but a call
    // that failed half-way would not): clear the part this call uses.
    const size_t n_elems = size_t(width) * height * 3;
    unsigned long long* fx = sp + PT_SCRATCH_HEAD_WORDS;
    PT_HIP(h, hipMemsetAsync(fx, 0, n_elems * sizeof(unsigned long long), h->stream));
    if (int rc = render_core(h, width, height, spp_begin, spp_end, max_depth, seed, accum_rgb_f32, fx, sp, PT_SCRATCH_TAIL_ENTRIES, false)) return rc;
    if (ray_count_dev) PT_HIP(h, hipMemcpyAsync(ray_count_dev, sp, sizeof(unsigned long long), hipMemcpyDeviceToDevice, h->stream));
    return 0;
}

int pt_ray_count(pt_handle h, uint64_t* out) {
    if (!h || !out) return PT_ERR_INVALID;
    PT_HIP(h, hipSetDevice(h->device));
    unsigned long long v = 0;
    PT_HIP(h, hipMemcpyAsync(&v, h->d_counters + 3, sizeof(v), hipMemcpyDeviceToHost, h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    *out = v;
    return 0;
}

int pt_kernel_times_split(pt_handle h, float* out_ms, float* tail_ms, int cap, int* n) {
    if (!h || !n || (cap > 0 && !out_ms)) return PT_ERR_INVALID;
    PT_HIP(h, hipSetDevice(h->device));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    const int count = int(h->ev_used / 3);
    for (int i = 0; i < count && i < cap; ++i) {
        PT_HIP(h, hipEventElapsedTime(&out_ms[i], h->ev[3 * i], h->ev[3 * i + 2]));
        if (tail_ms) PT_HIP(h, hipEventElapsedTime(&tail_ms[i], h->ev[3 * i + 1], h->ev[3 * i + 2]));
    }
    *n = count;
    h->ev_used = 0;
    return 0;
}

int pt_kernel_times(pt_handle h, float* out_ms, int cap, int* n) { return pt_kernel_times_split(h, out_ms, nullptr, cap, n); }

// ---- RCCL reduce of the own accumulation buffers, single process, one handle per GPU ----
// librccl is loaded once and the communicators are kept for the device list they were made for (set-up costs
// hundreds of milliseconds, a reduce of a 1080p frame well under one); they are destroyed when the device list
// changes and when the last handle of the process is destroyed.
}   // extern "C"

namespace {
typedef void* comm_t;
struct RcclState {
    std::mutex mutex;
    void* lib = nullptr;
    int (*init_all)(comm_t*, int, const int*) = nullptr;
    int (*reduce)(const void*, void*, size_t, int, int, int, comm_t, hipStream_t) = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    int (*comm_destroy)(comm_t) = nullptr;
    int (*comm_abort)(comm_t) = nullptr;
    std::vector<int> devices;          // device of each rank of the cached communicators
    std::vector<comm_t> comms;
    int inits = 0;                     // ncclCommInitAll calls made so far (pt_reduce_stats)
    int live_handles = 0;
    void drop(bool abort) {            // mutex held
        for (comm_t c : comms) if (c) (void)((abort && comm_abort) ? comm_abort(c) : comm_destroy(c));
        comms.clear(); devices.clear();
    }
};
RcclState g_rccl;
}   // namespace

static void rccl_handle_created() { std::lock_guard<std::mutex> lock(g_rccl.mutex); ++g_rccl.live_handles; }
static void rccl_handle_destroyed() {
    std::lock_guard<std::mutex> lock(g_rccl.mutex);
    if (--g_rccl.live_handles <= 0 && !g_rccl.comms.empty()) g_rccl.drop(false);
}

extern "C" {

int pt_reduce(pt_handle* handles, int n, int root) {
    if (!handles || n <= 0 || root < 0 || root >= n) return PT_ERR_INVALID;
    for (int i = 0; i < n; ++i) if (!handles[i]) return PT_ERR_INVALID;
    pt_handle h0 = handles[0];
    for (int i = 0; i < n; ++i) {
        if (!handles[i]->d_accum) return fail(h0, PT_ERR_STATE, "pt_reduce: a handle has no accumulation buffer");
        if (handles[i]->acc_w != h0->acc_w || handles[i]->acc_h != h0->acc_h)
            return fail(h0, PT_ERR_INVALID, "pt_reduce: accumulation buffers differ in size");
        for (int k = 0; k < i; ++k)
            if (handles[k]->device == handles[i]->device)
                return fail(h0, PT_ERR_INVALID, "pt_reduce: two handles on device " + std::to_string(handles[i]->device) +
                                                    " (one handle per GPU: RCCL takes one rank per device)");
    }
    // One handle: nothing to sum. (PT_REDUCE_FORCE_RCCL=1 goes through RCCL all the same, a one-rank communicator and an
    // in-place reduce that leaves the buffer as it is: the only way to run this function's RCCL calls on a one-GPU machine.)
    if (n == 1 && !(std::getenv("PT_REDUCE_FORCE_RCCL") && std::atoi(std::getenv("PT_REDUCE_FORCE_RCCL")) != 0)) return 0;
    std::lock_guard<std::mutex> lock(g_rccl.mutex);
    RcclState& R = g_rccl;
    if (!R.lib) {
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return fail(h0, PT_ERR_RCCL, std::string("pt_reduce: cannot load librccl: ") + dlerror());
        R.init_all = reinterpret_cast<decltype(R.init_all)>(dlsym(lib, "ncclCommInitAll"));
        R.reduce = reinterpret_cast<decltype(R.reduce)>(dlsym(lib, "ncclReduce"));
        R.group_start = reinterpret_cast<decltype(R.group_start)>(dlsym(lib, "ncclGroupStart"));
        R.group_end = reinterpret_cast<decltype(R.group_end)>(dlsym(lib, "ncclGroupEnd"));
        R.comm_destroy = reinterpret_cast<decltype(R.comm_destroy)>(dlsym(lib, "ncclCommDestroy"));
        R.comm_abort = reinterpret_cast<decltype(R.comm_abort)>(dlsym(lib, "ncclCommAbort"));      // optional
        if (!R.init_all || !R.reduce || !R.group_start || !R.group_end || !R.comm_destroy) {
            (void)dlclose(lib);
            return fail(h0, PT_ERR_RCCL, "pt_reduce: librccl lacks an expected symbol");
        }
        R.lib = lib;
    }
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) devs[i] = handles[i]->device;
    if (devs != R.devices) {
        R.drop(false);
        std::vector<comm_t> comms(n, nullptr);
        ++R.inits;
        if (R.init_all(comms.data(), n, devs.data()) != 0) return fail(h0, PT_ERR_RCCL, "pt_reduce: ncclCommInitAll failed");
        R.comms = comms; R.devices = devs;
    }
    // The reduce runs on each handle's stream, after that handle's renders; each handle is idle otherwise
    // (one host thread per handle at a time, as everywhere in this library).
    const size_t count = size_t(h0->acc_w) * h0->acc_h * 3;
    const int ncclFloat32 = 7, ncclSum = 0;
    int bad = 0;
    if (R.group_start() != 0) return fail(h0, PT_ERR_RCCL, "pt_reduce: ncclGroupStart failed");
    for (int i = 0; i < n && !bad; ++i) {
        (void)hipSetDevice(handles[i]->device);
        if (R.reduce(handles[i]->d_accum, handles[i]->d_accum, count, ncclFloat32, ncclSum, root, R.comms[i], handles[i]->stream) != 0) bad = 1;
    }
    if (R.group_end() != 0) bad = 1;
    if (bad) {
        R.drop(true);                                      // abort rather than wait for a collective that will not complete
        return fail(h0, PT_ERR_RCCL, "pt_reduce: ncclReduce failed");
    }
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(handles[i]->device);
        hipError_t e = hipStreamSynchronize(handles[i]->stream);
        if (e != hipSuccess) { R.drop(true); return fail(h0, PT_ERR_HIP, std::string("pt_reduce: hipStreamSynchronize: ") + hipGetErrorString(e)); }
    }
    return 0;
}

int pt_reduce_stats(int* comm_inits, int* cached_ranks) {
    std::lock_guard<std::mutex> lock(g_rccl.mutex);
    if (comm_inits) *comm_inits = g_rccl.inits;
    if (cached_ranks) *cached_ranks = int(g_rccl.comms.size());
    return 0;
}

// ---- unit-test entry points ----
// Start of a unit-test call: room for `bytes` in the arena (the previous call ended with a stream synchronise).
static int arena_begin(pt_handle h, size_t bytes) {
    if (bytes > h->ut_cap) {
        if (h->d_ut) { PT_HIP(h, hipFree(h->d_ut)); h->d_ut = nullptr; h->ut_cap = 0; }
        const size_t cap = std::max<size_t>(bytes, 1 << 16);
        PT_HIP(h, hipMalloc(reinterpret_cast<void**>(&h->d_ut), cap));
        h->ut_cap = cap;
    }
    h->ut_used = 0;
    return 0;
}
static int arena_take(pt_handle h, DevBuf& b, size_t bytes) {
    const size_t need = (bytes + 255) & ~size_t(255);
    if (h->ut_used + need > h->ut_cap) return fail(h, PT_ERR_STATE, "unit-test arena too small (internal)");
    b.p = h->d_ut + h->ut_used;
    h->ut_used += need;
    return 0;
}
#define PT_ARENA(bytes_per_element, buffers) do { if (int rc = arena_begin(h, size_t(n) * (bytes_per_element) + (buffers) * 256)) return rc; } while (0)
#define PT_ALLOC_COPY(buf, ptr, bytes)                                                              \
    do {                                                                                            \
        if (int rc = arena_take(h, buf, bytes)) return rc;                                          \
        if (ptr) PT_HIP(h, hipMemcpyAsync(buf.p, ptr, bytes, hipMemcpyHostToDevice, h->stream));    \
    } while (0)
#define PT_COPY_BACK(ptr, buf, bytes) PT_HIP(h, hipMemcpyAsync(ptr, buf.p, bytes, hipMemcpyDeviceToHost, h->stream))

int pt_scan(pt_handle h, int n, const double* o, const double* d, double t_min, double* t, int* idx, double* p,
            double* nrm, int* front) {
    if (!h) return PT_ERR_INVALID;
    if (!h->have_scene) return fail(h, PT_ERR_STATE, "pt_scan: set the scene first");
    if (n < 0 || (n > 0 && (!o || !d || !t || !idx || !p || !nrm || !front))) return fail(h, PT_ERR_INVALID, "pt_scan: bad arguments");
    if (n == 0) return 0;
    PT_HIP(h, hipSetDevice(h->device));
    DevBuf bo, bd, bt, bi, bp, bn, bf;
    PT_ARENA(4 * 24 + 8 + 4 + 4, 7);
    const size_t v3 = size_t(n) * 3 * sizeof(double);
    PT_ALLOC_COPY(bo, o, v3); PT_ALLOC_COPY(bd, d, v3);
    PT_ALLOC_COPY(bt, (void*)nullptr, n * sizeof(double)); PT_ALLOC_COPY(bi, (void*)nullptr, n * sizeof(int));
    PT_ALLOC_COPY(bp, (void*)nullptr, v3); PT_ALLOC_COPY(bn, (void*)nullptr, v3); PT_ALLOC_COPY(bf, (void*)nullptr, n * sizeof(int));
    if (h->accel == PT_ACCEL_GRID) {
        if (!h->grid_ready) return fail(h, PT_ERR_STATE, "pt_scan: the grid has not been built");
        if (h->grid_info[1] == 1)
            hipLaunchKernelGGL(k_scan<2>, dim3((n + 63) / 64), dim3(64), 0, h->stream, scene_dev(h), h->grid, n, bo.as<double>(), bd.as<double>(),
                               t_min, bt.as<double>(), bi.as<int>(), bp.as<double>(), bn.as<double>(), bf.as<int>());
        else
            hipLaunchKernelGGL(k_scan<1>, dim3((n + 63) / 64), dim3(64), 0, h->stream, scene_dev(h), h->grid, n, bo.as<double>(), bd.as<double>(),
                               t_min, bt.as<double>(), bi.as<int>(), bp.as<double>(), bn.as<double>(), bf.as<int>());
    } else {
        GridDev none;
        std::memset(&none, 0, sizeof(none));
        hipLaunchKernelGGL(k_scan<0>, dim3((n + 63) / 64), dim3(64), 0, h->stream, scene_dev(h), none, n, bo.as<double>(), bd.as<double>(),
                           t_min, bt.as<double>(), bi.as<int>(), bp.as<double>(), bn.as<double>(), bf.as<int>());
    }
    PT_HIP(h, hipGetLastError());
    PT_COPY_BACK(t, bt, n * sizeof(double)); PT_COPY_BACK(idx, bi, n * sizeof(int));
    PT_COPY_BACK(p, bp, v3); PT_COPY_BACK(nrm, bn, v3); PT_COPY_BACK(front, bf, n * sizeof(int));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pt_trace(pt_handle h, int n, const double* o, const double* d, int max_depth, const double* u, double* rgb, int* n_rays) {
    if (!h) return PT_ERR_INVALID;
    if (!h->have_scene) return fail(h, PT_ERR_STATE, "pt_trace: set the scene first");
    if (n < 0 || max_depth < 0 || max_depth > 4096 || (n > 0 && (!o || !d || !rgb || !n_rays || (max_depth > 0 && !u))))
        return fail(h, PT_ERR_INVALID, "pt_trace: bad arguments");
    if (n == 0) return 0;
    PT_HIP(h, hipSetDevice(h->device));
    DevBuf bo, bd, bu, bc, br;
    const size_t v3 = size_t(n) * 3 * sizeof(double), ub = std::max<size_t>(size_t(n) * size_t(max_depth) * 3 * sizeof(double), 8);
    if (int rc = arena_begin(h, 3 * v3 + ub + size_t(n) * sizeof(int) + 5 * 256)) return rc;
    PT_ALLOC_COPY(bo, o, v3); PT_ALLOC_COPY(bd, d, v3);
    PT_ALLOC_COPY(bu, max_depth > 0 ? u : nullptr, ub);
    PT_ALLOC_COPY(bc, (void*)nullptr, v3); PT_ALLOC_COPY(br, (void*)nullptr, n * sizeof(int));
    const dim3 grid_dim((n + 63) / 64), block(64);
    if (h->accel == PT_ACCEL_GRID) {
        if (!h->grid_ready) return fail(h, PT_ERR_STATE, "pt_trace: the grid has not been built");
        if (h->grid_info[1] == 1)
            hipLaunchKernelGGL(k_trace<2>, grid_dim, block, 0, h->stream, scene_dev(h), h->grid, n, bo.as<double>(), bd.as<double>(), max_depth,
                               bu.as<double>(), bc.as<double>(), br.as<int>());
        else
            hipLaunchKernelGGL(k_trace<1>, grid_dim, block, 0, h->stream, scene_dev(h), h->grid, n, bo.as<double>(), bd.as<double>(), max_depth,
                               bu.as<double>(), bc.as<double>(), br.as<int>());
    } else {
        GridDev none;
        std::memset(&none, 0, sizeof(none));
        hipLaunchKernelGGL(k_trace<0>, grid_dim, block, 0, h->stream, scene_dev(h), none, n, bo.as<double>(), bd.as<double>(), max_depth,
                           bu.as<double>(), bc.as<double>(), br.as<int>());
    }
    PT_HIP(h, hipGetLastError());
    PT_COPY_BACK(rgb, bc, v3); PT_COPY_BACK(n_rays, br, n * sizeof(int));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pt_scatter(pt_handle h, int n, const int* kind, const double* albedo, const double* fuzz, const double* ior,
               const double* d_in, const double* p, const double* nrm, const int* front, const double* u,
               double* d_out, double* atten, int* ok) {
    (void)p;   // the scattered ray starts at the hit point unchanged (Lambertian.cpp:13, Metal.cpp:10, Dielectric.cpp:41)
    if (!h) return PT_ERR_INVALID;
    if (n < 0 || (n > 0 && (!kind || !albedo || !fuzz || !ior || !d_in || !nrm || !front || !u || !d_out || !atten || !ok)))
        return fail(h, PT_ERR_INVALID, "pt_scatter: bad arguments");
    if (n == 0) return 0;
    for (int i = 0; i < n; ++i) if (kind[i] < 0 || kind[i] > 2) return fail(h, PT_ERR_INVALID, "pt_scatter: unknown material kind");
    PT_HIP(h, hipSetDevice(h->device));
    DevBuf bk, ba, bfz, bio, bd, bn, bfr, bu, bdo, bat, bok;
    PT_ARENA(6 * 24 + 2 * 8 + 3 * 4, 11);
    const size_t v3 = size_t(n) * 3 * sizeof(double), s1 = size_t(n) * sizeof(double), i1 = size_t(n) * sizeof(int);
    PT_ALLOC_COPY(bk, kind, i1); PT_ALLOC_COPY(ba, albedo, v3); PT_ALLOC_COPY(bfz, fuzz, s1); PT_ALLOC_COPY(bio, ior, s1);
    PT_ALLOC_COPY(bd, d_in, v3); PT_ALLOC_COPY(bn, nrm, v3); PT_ALLOC_COPY(bfr, front, i1); PT_ALLOC_COPY(bu, u, v3);
    PT_ALLOC_COPY(bdo, (void*)nullptr, v3); PT_ALLOC_COPY(bat, (void*)nullptr, v3); PT_ALLOC_COPY(bok, (void*)nullptr, i1);
    hipLaunchKernelGGL(k_scatter, dim3((n + 63) / 64), dim3(64), 0, h->stream, n, bk.as<int>(), ba.as<double>(), bfz.as<double>(),
                       bio.as<double>(), bd.as<double>(), bn.as<double>(), bfr.as<int>(), bu.as<double>(), bdo.as<double>(),
                       bat.as<double>(), bok.as<int>());
    PT_HIP(h, hipGetLastError());
    PT_COPY_BACK(d_out, bdo, v3); PT_COPY_BACK(atten, bat, v3); PT_COPY_BACK(ok, bok, i1);
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pt_gen_ray(pt_handle h, int n, int width, int height, const int* pi, const int* pj, const double* u1,
               const double* u2, const double* diskx, const double* disky, double* o, double* d) {
    if (!h) return PT_ERR_INVALID;
    if (!h->have_camera) return fail(h, PT_ERR_STATE, "pt_gen_ray: set the camera first");
    if (n < 0 || width <= 0 || height <= 0 || (n > 0 && (!pi || !pj || !u1 || !u2 || !diskx || !disky || !o || !d)))
        return fail(h, PT_ERR_INVALID, "pt_gen_ray: bad arguments");
    if (n == 0) return 0;
    PT_HIP(h, hipSetDevice(h->device));
    DevBuf bi, bj, b1, b2, bx, by, bo, bd;
    PT_ARENA(2 * 24 + 4 * 8 + 2 * 4, 8);
    const size_t v3 = size_t(n) * 3 * sizeof(double), s1 = size_t(n) * sizeof(double), i1 = size_t(n) * sizeof(int);
    PT_ALLOC_COPY(bi, pi, i1); PT_ALLOC_COPY(bj, pj, i1); PT_ALLOC_COPY(b1, u1, s1); PT_ALLOC_COPY(b2, u2, s1);
    PT_ALLOC_COPY(bx, diskx, s1); PT_ALLOC_COPY(by, disky, s1);
    PT_ALLOC_COPY(bo, (void*)nullptr, v3); PT_ALLOC_COPY(bd, (void*)nullptr, v3);
    hipLaunchKernelGGL(k_gen_ray, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->cam, n, 1.0f / float(width - 1),
                       1.0f / float(height - 1), bi.as<int>(), bj.as<int>(), b1.as<double>(), b2.as<double>(), bx.as<double>(),
                       by.as<double>(), bo.as<double>(), bd.as<double>());
    PT_HIP(h, hipGetLastError());
    PT_COPY_BACK(o, bo, v3); PT_COPY_BACK(d, bd, v3);
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pt_sky(pt_handle h, int n, const double* d, double* rgb) {
    if (!h) return PT_ERR_INVALID;
    if (n < 0 || (n > 0 && (!d || !rgb))) return fail(h, PT_ERR_INVALID, "pt_sky: bad arguments");
    if (n == 0) return 0;
    PT_HIP(h, hipSetDevice(h->device));
    DevBuf bd, bc;
    PT_ARENA(2 * 24, 2);
    const size_t v3 = size_t(n) * 3 * sizeof(double);
    PT_ALLOC_COPY(bd, d, v3); PT_ALLOC_COPY(bc, (void*)nullptr, v3);
    hipLaunchKernelGGL(k_sky, dim3((n + 63) / 64), dim3(64), 0, h->stream, n, bd.as<double>(), bc.as<double>());
    PT_HIP(h, hipGetLastError());
    PT_COPY_BACK(rgb, bc, v3);
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

int pt_philox(pt_handle h, int n, const uint32_t* ctr, const uint32_t key[2], uint32_t* out) {
    if (!h) return PT_ERR_INVALID;
    if (n < 0 || !key || (n > 0 && (!ctr || !out))) return fail(h, PT_ERR_INVALID, "pt_philox: bad arguments");
    if (n == 0) return 0;
    PT_HIP(h, hipSetDevice(h->device));
    DevBuf bc, bo;
    PT_ARENA(2 * 16, 2);
    const size_t b = size_t(n) * 4 * sizeof(uint32_t);
    PT_ALLOC_COPY(bc, ctr, b); PT_ALLOC_COPY(bo, (void*)nullptr, b);
    hipLaunchKernelGGL(k_philox, dim3((n + 63) / 64), dim3(64), 0, h->stream, n, bc.as<uint32_t>(), key[0], key[1], bo.as<uint32_t>());
    PT_HIP(h, hipGetLastError());
    PT_COPY_BACK(out, bo, b);
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return 0;
}

}   // extern "C"

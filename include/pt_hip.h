/* pt_hip.h — C-ABI of the MI355X (gfx950) sample-loop library, libpt_hip.so.
 *
 * The reference (DryPerspective/Path-Tracer) is one executable with no plugin, operator or FFI
 * boundary: its hot path is reached through the closure `sumColour(i, j, raysToCalc)`
 * (Path Tracer.cpp:265-278), called once per pixel by the row task (Path Tracer.cpp:296-300).
 * This header is the seam a host program binds instead of that closure. Each entry point cites
 * the reference interface it replaces.
 *
 * Rules of the boundary
 *  - plain C types only; every function returns 0 on success and a non-zero code otherwise,
 *    with the text available from pt_last_error(); no C++ exception crosses the boundary;
 *  - a handle is used by one host thread at a time; different handles (one per GPU) may be used
 *    from different threads;
 *  - pt_render() is asynchronous on the handle's stream unless it is asked for the ray count;
 *    pt_read_accum(), pt_ray_count(), pt_kernel_times() and pt_synchronize() synchronise;
 *  - there is no CPU fallback: without a usable gfx950 device pt_create() fails.
 *
 * Scene arrays are in LIST ORDER (ties between equal hit distances go to the later index,
 * HittableList.cpp:27-32 with Sphere.cpp:18). Host-side values are doubles, as in the reference;
 * the library keeps doubles for the hit record of the winning sphere and floats for the scan.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_context* pt_handle;

/* Material kinds, in the order of the reference's classes (Lambertian.h, Metal.h, Dielectric.h). */
enum { PT_LAMBERTIAN = 0, PT_METAL = 1, PT_DIELECTRIC = 2 };

/* Error codes (besides 0). */
enum {
    PT_ERR_INVALID = 1,   /* bad argument (null pointer, non-positive size, unknown kind ...) */
    PT_ERR_HIP = 2,       /* a HIP runtime call failed; pt_last_error() has hipGetErrorString() */
    PT_ERR_STATE = 3,     /* call out of order (render before set_scene / set_camera ...) */
    PT_ERR_RCCL = 4       /* an RCCL call failed */
};

/* Tuning variables, read from the environment ONCE, in pt_create (a handle keeps what it read). None changes a result:
 * the sums are integers, a frame is bit-identical for any job shape (tests/test_gpu_parity.py renders one frame under six
 * settings). Defaults are what the committed measurements used; they exist for the A/B runs of tools/.
 *   PT_TARGET_JOBS   (262144) jobs a frame is cut into at least (job = 8x8 pixel tile x a chunk of samples)
 *   PT_MIN_CHUNK     (3)      fewest samples per pixel in a job
 *   PT_FINE_CHUNK    (2)      samples per pixel of the small jobs handed out last (0: none)
 *   PT_FINE_SHARE    (100)    their total, in percent of one large job per resident wave
 *   PT_TAIL_ROUNDS   (1)      workgroups of the end-of-frame kernel, in units of what the chip holds at once
 *   PT_GRID_PCT      (100)    persistent grid size in percent of the resident waves
 *   PT_GRID_CELL_SCALE (4.0), PT_GRID_BIG_RATIO (3.0), PT_GRID_FLAT (1)   shape of the optional grid search (pt_set_accel)
 * A production host that must not depend on the environment clears these variables before pt_create. */

/* One context per GPU. Replaces the construction of dp::thread_pool (Path Tracer.cpp:286,
 * ThreadPool.h:132-144): the pool of hardware_concurrency() workers becomes one device. */
int pt_create(int device_index, pt_handle* out);
int pt_destroy(pt_handle h);

/* Text of the last error on this handle (h may be NULL: last error of pt_create on this thread). */
const char* pt_last_error(pt_handle h);

/* Run on a caller-owned hipStream_t (for example torch's current stream). NULL = the handle's own. */
int pt_set_stream(pt_handle h, void* hip_stream);
int pt_synchronize(pt_handle h);

/* Replaces HittableList::add of Sphere objects with their materials (HittableList.cpp:5-7,
 * Sphere.h:25, Lambertian.h:20, Metal.h:21, Dielectric.h:23), flattened by the host in list order.
 * kind[i] is PT_LAMBERTIAN / PT_METAL / PT_DIELECTRIC; albedo is used by the first two, fuzz by
 * Metal (already clamped to <= 1, Metal.h:21), ior by Dielectric. Arrays are read during the call.
 * Rejected with PT_ERR_INVALID: a non-finite centre, a radius <= 0, an unknown kind, a Dielectric with ior <= 0, an
 * albedo component that is negative or not finite (the order-independent fixed-point sums of pt_render hold
 * non-negative radiance only). */
int pt_set_scene(pt_handle h, int n, const double* cx, const double* cy, const double* cz, const double* r,
                 const int* kind, const double* albedo_r, const double* albedo_g, const double* albedo_b,
                 const double* fuzz, const double* ior);

/* Optional: how pt_render and pt_scan look for the closest hit. PT_ACCEL_NONE (the default) tests every
 * sphere for every ray, as HittableList::isHit does (HittableList.cpp:27). PT_ACCEL_GRID walks a uniform
 * grid built from the scene and tests only the spheres entered in the cells the ray crosses; the hit
 * returned obeys the same rule (closest t in range, ties to the later list index). Meant for scenes of
 * thousands of spheres; its throughput is not counted in the FP32 roofline figure, whose unit is one
 * ray-sphere test of the full scan. May be called before or after pt_set_scene. */
enum { PT_ACCEL_NONE = 0, PT_ACCEL_GRID = 1 };
int pt_set_accel(pt_handle h, int mode);
/* Shape of the grid in use: info = nx, ny, nz, spheres kept in the linearly scanned "big" list, entries
 * in all cells, entries in the fullest cell; *cell_edge (may be NULL) the cell size. */
int pt_accel_info(pt_handle h, int info[6], double* cell_edge);

/* The seven values Camera::setupCamera derives (Camera.cpp:27-31) and Camera::getCurrentRay reads
 * (Camera.cpp:38-46): position, lower-left corner, horizontal, vertical, unit u, unit v, lens radius. */
int pt_set_camera(pt_handle h, const double position[3], const double lower_left[3], const double horizontal[3],
                  const double vertical[3], const double u[3], const double v[3], double lens_radius);

/* Replaces the row tasks and sumColour (Path Tracer.cpp:265-278, :296-305): adds the radiance of
 * samples [spp_begin, spp_end) of EVERY pixel into accum_rgb_f32, a DEVICE buffer of
 * width*height*3 floats, top row first (the reference's `pixels` vector, Path Tracer.cpp:292-298).
 * accum_rgb_f32 == NULL selects the handle's own buffer (see pt_clear_accum / pt_read_accum).
 * Random numbers are philox4x32-10 blocks keyed by `seed` and numbered (pixel, sample, bounce),
 * so disjoint spp ranges of one frame can be rendered on different GPUs and summed.
 * ray_count_out, when not NULL, makes the call synchronous and receives the number of rays
 * traced by this call (calcColour entries with depth > 0, Path Tracer.cpp:82-92). */
int pt_render(pt_handle h, int width, int height, int spp_begin, int spp_end, int max_depth, uint64_t seed,
              float* accum_rgb_f32, uint64_t* ray_count_out);

/* pt_render for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc). No counterpart in the reference. Same arguments and the same
 * kernels as pt_render, bit-identical sums, with three differences:
 *  - the job sums, the counters and the hand-over slots of the end-of-frame kernel, which pt_render keeps in
 *    buffers it allocates, live in a fixed array inside the library (static __device__: 3 x 8 bytes per pixel
 *    of the largest image + 8.4 MB = 58.2 MB on every GPU the library runs on, held until the process ends):
 *    width*height <= PT_NOALLOC_MAX_PIXELS (1920x1080), larger images are rejected with PT_ERR_INVALID;
 *  - accum_rgb_f32 must be the caller's DEVICE buffer (NULL is rejected); ray_count_dev, when not NULL, is
 *    a DEVICE pointer that receives the rays traced by this call, written on the handle's stream; the call
 *    never synchronises and records no kernel time (pt_kernel_times does not see it);
 *  - every call on one GPU, from any handle, shares that GPU's array: calls that overlap on one GPU race.
 *    Run them one after the other (one stream, or ordered by events).
 * It calls no hipMalloc / hipFree / allocating routine; pt_create did the set-up (the array's address on the
 * handle's device, the occupancy query). */
#define PT_NOALLOC_MAX_PIXELS (1920 * 1080)
int pt_render_noalloc(pt_handle h, int width, int height, int spp_begin, int spp_end, int max_depth, uint64_t seed,
                      float* accum_rgb_f32, uint64_t* ray_count_dev);

/* The handle's own accumulation buffer (zero-filled, width*height*3 floats). */
int pt_clear_accum(pt_handle h, int width, int height);
/* Copies the handle's buffer to host_f32 (replaces reading `pixels` at Path Tracer.cpp:320). width and height
 * state the size of the caller's array: PT_ERR_INVALID unless they are the buffer's own. */
int pt_read_accum(pt_handle h, int width, int height, float* host_f32);
void* pt_accum_device_ptr(pt_handle h);

/* Upload a saved accumulation buffer (width*height*3 floats, top row first) into the handle's own
 * buffer, so that a frame can be resumed: write, then pt_render() further sample ranges. The reference
 * keeps its image in memory until the end (Path Tracer.cpp:292-293, :316) and cannot resume. */
int pt_write_accum(pt_handle h, int width, int height, const float* host_f32);

/* The output stage on the device: gamma, clamp and quantise (writeColour, Path Tracer.cpp:62-74) of
 * the sums over `spp` samples in accum_rgb_f32 (DEVICE buffer; NULL = the handle's own), in the
 * reference's double arithmetic, into width*height*3 bytes at host_rgb8 (top row first, r g b).
 * The bytes are the numbers the reference prints into its P3 file. Synchronises. */
int pt_quantise_rgb8(pt_handle h, int width, int height, int spp, const float* accum_rgb_f32, uint8_t* host_rgb8);

/* Rays traced since the handle was created (synchronises). */
int pt_ray_count(pt_handle h, uint64_t* out);

/* Durations, in milliseconds, of the sample-loop kernel of each pt_render() since the last call
 * to this function, measured with HIP events on the stream the kernel ran on. Synchronises.
 * Writes at most cap values and the number of renders recorded into *n. */
int pt_kernel_times(pt_handle h, float* out_ms, int cap, int* n);
/* The same records with the part spent in the end-of-frame kernel (k_tail, launched behind the sample-loop kernel to
 * finish the few long paths its waves were left with; 0 for the grid search) in tail_ms (may be NULL): out_ms[i] is the
 * whole, out_ms[i] - tail_ms[i] the sample-loop kernel alone. Like pt_kernel_times it clears the records. */
int pt_kernel_times_split(pt_handle h, float* out_ms, float* tail_ms, int cap, int* n);

/* Sum the own accumulation buffers of n handles (one per GPU of this process) into handles[root]
 * with an RCCL reduce. No counterpart in the reference, which shares one vector between threads
 * (Path Tracer.cpp:292); comes from the spp split of the frame across GPUs. */
int pt_reduce(pt_handle* handles, int n, int root);
/* The RCCL library and the communicators are set up on the first pt_reduce and kept for the same device list
 * (destroyed when the list changes or the last handle goes). One handle per device; two handles on one device
 * are rejected. comm_inits = communicator set-ups made so far, cached_ranks = ranks of the set in use.
 * (Diagnostic: with PT_REDUCE_FORCE_RCCL=1 in the environment a ONE-handle pt_reduce, which otherwise returns at once,
 * goes through a one-rank RCCL communicator: the way to run these calls on a machine with one GPU.) */
int pt_reduce_stats(int* comm_inits, int* cached_ranks);

/* ---- Unit-test entry points: the device functions of the sample loop, one call per batch. ---- */

/* HittableList::isHit over the scene set with pt_set_scene (HittableList.cpp:21-36, Sphere.cpp:5-33,
 * Hittable.h:26-31). o, d are [n][3] ray origins and (un-normalised) directions; t is in units of
 * |d| like the reference's; idx is the list index or -1. p, nrm are [n][3]. */
int pt_scan(pt_handle h, int n, const double* o, const double* d, double t_min, double* t, int* idx, double* p,
            double* nrm, int* front);

/* calcColour (Path Tracer.cpp:82-108) for n given rays, with the uniform numbers of every scatter injected instead of
 * drawn: the loop of pt_render (closest hit -> sky on a miss, Path Tracer.cpp:105-107 | hit record, depth rule :86,
 * Material::isScattered :96, product of attenuations :97, refused scatter -> black :100) on the same device functions,
 * without philox, regeneration and sums, so that whole paths can be pinned by hand-derived vectors. o, d are [n][3]
 * (d un-normalised, like the reference's rays); u is [n][max_depth][3]: the scatter at the k-th hit of path i takes
 * u[i][k] = (u1, u2 for the sphere point, VectorFunc.cpp:40-41; u3 for the Fresnel coin, Dielectric.cpp:27).
 * rgb is [n][3]; n_rays[i] = rays traced for path i (calcColour entries with depth > 0). max_depth = 0: black, 0 rays. */
int pt_trace(pt_handle h, int n, const double* o, const double* d, int max_depth, const double* u, double* rgb, int* n_rays);

/* Material::isScattered with the uniform numbers injected (Lambertian.cpp:5-16, Metal.cpp:6-15,
 * Dielectric.cpp:5-50, VectorFunc.cpp:37-43, :66-76). albedo, d_in, p, nrm, u, d_out, atten are [n][3];
 * u[k] = (u1, u2) for the sphere point and u3 for the Fresnel coin. ok = the returned bool. */
int pt_scatter(pt_handle h, int n, const int* kind, const double* albedo, const double* fuzz, const double* ior,
               const double* d_in, const double* p, const double* nrm, const int* front, const double* u,
               double* d_out, double* atten, int* ok);

/* Path Tracer.cpp:270-271 + Camera::getCurrentRay (Camera.cpp:38-46) with the jitter (u1, u2) and the
 * unit-disk point injected; the camera is the one set with pt_set_camera. o, d are [n][3]. */
int pt_gen_ray(pt_handle h, int n, int width, int height, const int* pi, const int* pj, const double* u1,
               const double* u2, const double* diskx, const double* disky, double* o, double* d);

/* The background colour of a missed ray (Path Tracer.cpp:105-107). d, rgb are [n][3]. */
int pt_sky(pt_handle h, int n, const double* d, double* rgb);

/* philox4x32-10 blocks as the kernels draw them. ctr is [n][4], out is [n][4]. */
int pt_philox(pt_handle h, int n, const uint32_t* ctr, const uint32_t key[2], uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif /* PT_HIP_H */

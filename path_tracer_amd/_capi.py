"""ctypes binding of libpt_hip.so (include/pt_hip.h). No torch types cross this boundary.

There is no CPU path: if the library is missing, or no gfx950 device is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_HIP_LIB") or os.path.join(_HERE, "lib", "libpt_hip.so")   # PT_HIP_LIB: diagnostic builds

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint32)
_fp = C.POINTER(C.c_float)
_u64p = C.POINTER(C.c_uint64)

# name -> (restype, argtypes); every symbol include/pt_hip.h declares
SIGNATURES = {
    "pt_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pt_destroy": (C.c_int, [C.c_void_p]),
    "pt_last_error": (C.c_char_p, [C.c_void_p]),
    "pt_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pt_synchronize": (C.c_int, [C.c_void_p]),
    "pt_set_scene": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp]),
    "pt_set_accel": (C.c_int, [C.c_void_p, C.c_int]),
    "pt_accel_info": (C.c_int, [C.c_void_p, _ip, _dp]),
    "pt_set_camera": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double]),
    "pt_render": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_void_p, _u64p]),
    "pt_render_noalloc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]),
    "pt_clear_accum": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pt_read_accum": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp]),
    "pt_accum_device_ptr": (C.c_void_p, [C.c_void_p]),
    "pt_write_accum": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp]),
    "pt_quantise_rgb8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_uint8)]),
    "pt_ray_count": (C.c_int, [C.c_void_p, _u64p]),
    "pt_kernel_times": (C.c_int, [C.c_void_p, _fp, C.c_int, _ip]),
    "pt_kernel_times_split": (C.c_int, [C.c_void_p, _fp, _fp, C.c_int, _ip]),
    "pt_reduce": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "pt_reduce_stats": (C.c_int, [_ip, _ip]),
    "pt_scan": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.c_double, _dp, _ip, _dp, _dp, _ip]),
    "pt_trace": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _ip]),
    "pt_scatter": (C.c_int, [C.c_void_p, C.c_int, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp, _ip]),
    "pt_gen_ray": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, _dp, _dp]),
    "pt_sky": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp]),
    "pt_philox": (C.c_int, [C.c_void_p, C.c_int, _up, _up, _up]),
}


class PtError(RuntimeError):
    pass


_lib = None


def load_library():
    """Load libpt_hip.so and bind every declared symbol. Raises if the library was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtError("%s is missing: run `make` (or __graft_entry__.build()) first; there is no fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


class Device:
    """One pt_handle: the sample loop on one MI355X."""

    def __init__(self, device_index=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.pt_create(device_index, C.byref(h))
        if rc != 0:
            raise PtError("pt_create failed (%d): %s" % (rc, self.lib.pt_last_error(None).decode()))
        self.h = h
        self.n_spheres = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.pt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise PtError("%s failed (%d): %s" % (what, rc, self.lib.pt_last_error(self.h).decode()))

    # -- configuration -----------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._ck(self.lib.pt_set_stream(self.h, C.c_void_p(stream_ptr)), "pt_set_stream")

    def synchronize(self):
        self._ck(self.lib.pt_synchronize(self.h), "pt_synchronize")

    def set_scene(self, sc):
        """sc: dict of arrays cx, cy, cz, r, kind, ar, ag, ab, fuzz, ior in list order."""
        a = {k: _d(sc[k]) for k in ("cx", "cy", "cz", "r", "ar", "ag", "ab", "fuzz", "ior")}
        kind = _i(sc["kind"])
        n = len(kind)
        self._ck(self.lib.pt_set_scene(self.h, n, _p(a["cx"], _dp), _p(a["cy"], _dp), _p(a["cz"], _dp), _p(a["r"], _dp),
                                       _p(kind, _ip), _p(a["ar"], _dp), _p(a["ag"], _dp), _p(a["ab"], _dp),
                                       _p(a["fuzz"], _dp), _p(a["ior"], _dp)), "pt_set_scene")
        self.n_spheres = n

    def set_accel(self, mode):
        """0: test every sphere (default); 1: uniform grid (include/pt_hip.h, PT_ACCEL_GRID)."""
        self._ck(self.lib.pt_set_accel(self.h, int(mode)), "pt_set_accel")

    def accel_info(self):
        info = np.zeros(6, dtype=np.int32)
        cell = C.c_double(0)
        self._ck(self.lib.pt_accel_info(self.h, _p(info, _ip), C.byref(cell)), "pt_accel_info")
        return dict(dims=tuple(int(v) for v in info[:3]), big=int(info[3]), entries=int(info[4]), fullest=int(info[5]),
                    cell=cell.value)

    def set_camera(self, cam19):
        """cam19: position, lower_left, horizontal, vertical, u, v (3 each), lens radius."""
        c = _d(cam19)
        assert c.shape == (19,)
        v = [np.ascontiguousarray(c[3 * k:3 * k + 3]) for k in range(6)]
        self._ck(self.lib.pt_set_camera(self.h, *[_p(x, _dp) for x in v], float(c[18])), "pt_set_camera")

    # -- the sample loop ---------------------------------------------------------------------
    def render(self, width, height, spp_begin, spp_end, max_depth, seed, accum_ptr=None, want_rays=False):
        """Adds samples [spp_begin, spp_end) into the device buffer at accum_ptr (None: the handle's own)."""
        rays = C.c_uint64(0)
        self._ck(self.lib.pt_render(self.h, width, height, spp_begin, spp_end, max_depth, seed,
                                    C.c_void_p(accum_ptr) if accum_ptr else None,
                                    C.byref(rays) if want_rays else None), "pt_render")
        return int(rays.value) if want_rays else None

    def render_noalloc(self, width, height, spp_begin, spp_end, max_depth, seed, accum_ptr, ray_count_ptr=None):
        """pt_render for callers that cannot allocate: accum_ptr and ray_count_ptr are DEVICE pointers; asynchronous."""
        self._ck(self.lib.pt_render_noalloc(self.h, width, height, spp_begin, spp_end, max_depth, seed,
                                            C.c_void_p(accum_ptr) if accum_ptr else None,
                                            C.c_void_p(ray_count_ptr) if ray_count_ptr else None), "pt_render_noalloc")

    def clear_accum(self, width, height):
        self._ck(self.lib.pt_clear_accum(self.h, width, height), "pt_clear_accum")
        self._wh = (width, height)

    def read_accum(self, width, height):
        out = np.zeros((height, width, 3), dtype=np.float32)
        self._ck(self.lib.pt_read_accum(self.h, width, height, _p(out, _fp)), "pt_read_accum")
        return out

    def write_accum(self, sums):
        """Upload saved sums (H, W, 3 float32) into the handle's own buffer, to resume a frame."""
        a = np.ascontiguousarray(sums, dtype=np.float32)
        self._ck(self.lib.pt_write_accum(self.h, a.shape[1], a.shape[0], _p(a, _fp)), "pt_write_accum")

    def quantise_rgb8(self, width, height, spp, accum_ptr=None):
        """Gamma, clamp, quantise on the device (writeColour): returns uint8 (H, W, 3)."""
        out = np.zeros((height, width, 3), dtype=np.uint8)
        self._ck(self.lib.pt_quantise_rgb8(self.h, width, height, spp, C.c_void_p(accum_ptr) if accum_ptr else None,
                                           out.ctypes.data_as(C.POINTER(C.c_uint8))), "pt_quantise_rgb8")
        return out

    def render_image(self, width, height, spp, max_depth=50, seed=42, spp_begin=0):
        """Convenience: clear, render [spp_begin, spp_begin+spp), read back. Returns (sum[H,W,3] float32, rays)."""
        self.clear_accum(width, height)
        rays = self.render(width, height, spp_begin, spp_begin + spp, max_depth, seed, want_rays=True)
        return self.read_accum(width, height), rays

    def ray_count(self):
        v = C.c_uint64(0)
        self._ck(self.lib.pt_ray_count(self.h, C.byref(v)), "pt_ray_count")
        return int(v.value)

    def kernel_times(self, cap=8192):
        out = np.zeros(cap, dtype=np.float32)
        n = C.c_int(0)
        self._ck(self.lib.pt_kernel_times(self.h, _p(out, _fp), cap, C.byref(n)), "pt_kernel_times")
        return out[:min(n.value, cap)].copy()

    def kernel_times_split(self, cap=4096):
        """(whole, tail) kernel milliseconds per render since the last call: whole = k_render + k_tail, tail = k_tail alone."""
        out = np.zeros(cap, dtype=np.float32)
        tail = np.zeros(cap, dtype=np.float32)
        n = C.c_int(0)
        self._ck(self.lib.pt_kernel_times_split(self.h, _p(out, _fp), _p(tail, _fp), cap, C.byref(n)), "pt_kernel_times_split")
        k = min(n.value, cap)
        return out[:k].copy(), tail[:k].copy()

    # -- unit-test entry points ----------------------------------------------------------------
    def scan(self, o, d, t_min=0.001):
        o, d = _d(o), _d(d)
        n = o.shape[0]
        t = np.zeros(n)
        idx = np.zeros(n, dtype=np.int32)
        p = np.zeros((n, 3))
        nrm = np.zeros((n, 3))
        front = np.zeros(n, dtype=np.int32)
        self._ck(self.lib.pt_scan(self.h, n, _p(o, _dp), _p(d, _dp), t_min, _p(t, _dp), _p(idx, _ip), _p(p, _dp),
                                  _p(nrm, _dp), _p(front, _ip)), "pt_scan")
        return dict(t=t, idx=idx, p=p, n=nrm, front=front)

    def trace(self, o, d, max_depth, u):
        """calcColour for given rays with the uniforms of every scatter injected: u is [n][max_depth][3]."""
        o, d = _d(o), _d(d)
        n = o.shape[0]
        u = _d(np.asarray(u, dtype=np.float64).reshape(n, max(max_depth, 0), 3))
        rgb = np.zeros((n, 3))
        rays = np.zeros(n, dtype=np.int32)
        self._ck(self.lib.pt_trace(self.h, n, _p(o, _dp), _p(d, _dp), max_depth, _p(u, _dp) if u.size else None, _p(rgb, _dp),
                                   _p(rays, _ip)), "pt_trace")
        return dict(rgb=rgb, rays=rays)

    def scatter(self, kind, albedo, fuzz, ior, d_in, p, nrm, front, u):
        kind, front = _i(kind), _i(front)
        albedo, fuzz, ior, d_in, p, nrm, u = map(_d, (albedo, fuzz, ior, d_in, p, nrm, u))
        n = len(kind)
        d_out = np.zeros((n, 3))
        atten = np.zeros((n, 3))
        ok = np.zeros(n, dtype=np.int32)
        self._ck(self.lib.pt_scatter(self.h, n, _p(kind, _ip), _p(albedo, _dp), _p(fuzz, _dp), _p(ior, _dp),
                                     _p(d_in, _dp), _p(p, _dp), _p(nrm, _dp), _p(front, _ip), _p(u, _dp),
                                     _p(d_out, _dp), _p(atten, _dp), _p(ok, _ip)), "pt_scatter")
        return dict(d=d_out, atten=atten, ok=ok)

    def gen_ray(self, width, height, pi, pj, u1, u2, diskx, disky):
        pi, pj = _i(pi), _i(pj)
        u1, u2, diskx, disky = map(_d, (u1, u2, diskx, disky))
        n = len(pi)
        o = np.zeros((n, 3))
        d = np.zeros((n, 3))
        self._ck(self.lib.pt_gen_ray(self.h, n, width, height, _p(pi, _ip), _p(pj, _ip), _p(u1, _dp), _p(u2, _dp),
                                     _p(diskx, _dp), _p(disky, _dp), _p(o, _dp), _p(d, _dp)), "pt_gen_ray")
        return o, d

    def sky(self, d):
        d = _d(d)
        out = np.zeros_like(d)
        self._ck(self.lib.pt_sky(self.h, d.shape[0], _p(d, _dp), _p(out, _dp)), "pt_sky")
        return out

    def philox(self, ctr, key):
        ctr = np.ascontiguousarray(ctr, dtype=np.uint32).reshape(-1, 4)
        key = np.ascontiguousarray(key, dtype=np.uint32)
        out = np.zeros_like(ctr)
        self._ck(self.lib.pt_philox(self.h, ctr.shape[0], _p(ctr, _up), _p(key, _up), _p(out, _up)), "pt_philox")
        return out


def reduce(devices, root=0):
    """pt_reduce: sums the own accumulation buffers of `devices` (one Device per GPU, this process) into devices[root]."""
    lib = load_library()
    arr = (C.c_void_p * len(devices))(*[d.h for d in devices])
    rc = lib.pt_reduce(arr, len(devices), root)
    if rc != 0:
        raise PtError("pt_reduce failed (%d): %s" % (rc, lib.pt_last_error(devices[0].h).decode()))


def reduce_stats():
    """(communicator set-ups made so far, ranks of the cached set)."""
    lib = load_library()
    a, b = C.c_int(0), C.c_int(0)
    lib.pt_reduce_stats(C.byref(a), C.byref(b))
    return a.value, b.value

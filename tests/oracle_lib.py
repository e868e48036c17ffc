"""ctypes binding of the CPU oracle (oracle/oracle.cpp). Test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The oracle is a double-precision restatement of the reference's sample loop; see the header of
oracle/oracle.cpp for what pins it ("parity unpinned": the reference holds no fixtures and
does not build here).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "liboracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_uint32)


def build():
    """Compile the oracle with g++ (no-op when the .so is newer than the source)."""
    src = os.path.join(_ROOT, "oracle", "oracle.cpp")
    if os.path.exists(_SO) and os.path.getmtime(_SO) >= os.path.getmtime(src):
        return _SO
    subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


class Oracle:
    def __init__(self):
        if not os.path.exists(_SO):
            build()
        self.lib = C.CDLL(_SO)
        L = self.lib
        L.orc_mt19937_nth.restype = C.c_uint32
        L.orc_mt19937_nth.argtypes = [C.c_uint32, C.c_int]
        L.orc_uniform_draws.argtypes = [C.c_uint32, C.c_double, C.c_double, C.c_int, _dp]
        L.orc_philox4x32_10.argtypes = [_up, _up, _up]
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_generate_scene.restype = C.c_int
        L.orc_generate_scene.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double,
                                         _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp]
        L.orc_camera_setup.argtypes = [_dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _dp]
        L.orc_gen_ray.argtypes = [_dp, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_scan.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_double, _dp, _ip, _dp, _dp, _ip]
        L.orc_scatter.argtypes = [C.c_int, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp, _ip]
        L.orc_sky.argtypes = [C.c_int, _dp, _dp]
        L.orc_trace.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _ip]
        L.orc_render.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                 C.c_uint32, C.c_int, C.c_int, _dp, _dp, C.POINTER(C.c_uint64)]
        L.orc_format_ppm.restype = C.c_long
        L.orc_format_ppm.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_long]

    # -- RNG -------------------------------------------------------------------------------
    def mt19937_nth(self, seed, n):
        return int(self.lib.orc_mt19937_nth(seed, n))

    def uniform_draws(self, seed, a, b, n):
        out = np.zeros(n)
        self.lib.orc_uniform_draws(seed, a, b, n, out.ctypes.data_as(_dp))
        return out

    def philox(self, ctr, key):
        c = np.asarray(ctr, dtype=np.uint32)
        k = np.asarray(key, dtype=np.uint32)
        o = np.zeros(4, dtype=np.uint32)
        self.lib.orc_philox4x32_10(c.ctypes.data_as(_up), k.ctypes.data_as(_up), o.ctypes.data_as(_up))
        return o

    # -- scene / camera --------------------------------------------------------------------
    def generate_scene(self, count=250, seed_a=1, seed_b=2, half_extent=10.0, thr_diffuse=0.6, thr_metal=0.9):
        cap = count + 5
        f = {k: np.zeros(cap) for k in ("cx", "cy", "cz", "r", "ar", "ag", "ab", "fuzz", "ior")}
        kind = np.zeros(cap, dtype=np.int32)
        n = self.lib.orc_generate_scene(count, seed_a, seed_b, half_extent, thr_diffuse, thr_metal,
                                        *[f[k].ctypes.data_as(_dp) for k in ("cx", "cy", "cz", "r")],
                                        kind.ctypes.data_as(_ip),
                                        *[f[k].ctypes.data_as(_dp) for k in ("ar", "ag", "ab", "fuzz", "ior")])
        out = {k: v[:n].copy() for k, v in f.items()}
        out["kind"] = kind[:n].copy()
        return out

    def scene_create(self, sc):
        keep = [_d(sc[k]) for k in ("cx", "cy", "cz", "r")]
        kind = _i(sc["kind"])
        keep2 = [_d(sc[k]) for k in ("ar", "ag", "ab", "fuzz", "ior")]
        return self.lib.orc_scene_create(len(sc["cx"]), *[k[1] for k in keep], kind[1], *[k[1] for k in keep2])

    def scene_free(self, h):
        self.lib.orc_scene_free(h)

    def camera_setup(self, pos=(8, 2, 3), look=(0, 0, 0), up=(0, 1, 0), aspect=16.0 / 9.0, focal=1.0, fov=60.0,
                     aperture=0.1, focus=None):
        if focus is None:
            focus = float(np.sqrt(77.0))  # Path Tracer.cpp:127, frozen at the default camera
        p, pp = _d(pos)
        l, lp = _d(look)
        u, upp = _d(up)
        cam = np.zeros(19)
        self.lib.orc_camera_setup(pp, lp, upp, aspect, focal, fov, aperture, focus, cam.ctypes.data_as(_dp))
        return cam

    def gen_ray(self, cam, W, H, pi, pj, u1, u2, diskx, disky):
        n = len(pi)
        cam, cp = _d(cam)
        pi, pip = _i(pi)
        pj, pjp = _i(pj)
        u1, u1p = _d(u1)
        u2, u2p = _d(u2)
        dx, dxp = _d(diskx)
        dy, dyp = _d(disky)
        o = np.zeros((n, 3))
        d = np.zeros((n, 3))
        self.lib.orc_gen_ray(cp, n, W, H, pip, pjp, u1p, u2p, dxp, dyp, o.ctypes.data_as(_dp), d.ctypes.data_as(_dp))
        return o, d

    # -- per-ray -----------------------------------------------------------------------------
    def scan(self, scene, o, d, t_min=0.001):
        o, op = _d(o)
        d, dp = _d(d)
        n = o.shape[0]
        t = np.zeros(n)
        idx = np.zeros(n, dtype=np.int32)
        p = np.zeros((n, 3))
        nrm = np.zeros((n, 3))
        front = np.zeros(n, dtype=np.int32)
        self.lib.orc_scan(scene, n, op, dp, t_min, t.ctypes.data_as(_dp), idx.ctypes.data_as(_ip),
                          p.ctypes.data_as(_dp), nrm.ctypes.data_as(_dp), front.ctypes.data_as(_ip))
        return dict(t=t, idx=idx, p=p, n=nrm, front=front)

    def scatter(self, kind, albedo, fuzz, ior, d_in, p, nrm, front, u):
        n = len(kind)
        kind, kp = _i(kind)
        albedo, ap = _d(albedo)
        fuzz, fp = _d(fuzz)
        ior, ip_ = _d(ior)
        d_in, dp = _d(d_in)
        p, pp = _d(p)
        nrm, np_ = _d(nrm)
        front, frp = _i(front)
        u, up = _d(u)
        d_out = np.zeros((n, 3))
        atten = np.zeros((n, 3))
        ok = np.zeros(n, dtype=np.int32)
        self.lib.orc_scatter(n, kp, ap, fp, ip_, dp, pp, np_, frp, up, d_out.ctypes.data_as(_dp),
                             atten.ctypes.data_as(_dp), ok.ctypes.data_as(_ip))
        return dict(d=d_out, atten=atten, ok=ok)

    def sky(self, d):
        d, dp = _d(d)
        out = np.zeros_like(d)
        self.lib.orc_sky(d.shape[0], dp, out.ctypes.data_as(_dp))
        return out

    def trace(self, scene, o, d, max_depth, u):
        """calcColour for given rays, the uniforms of every scatter injected: u is [n][max_depth][3]."""
        o, op = _d(o)
        d, dp = _d(d)
        n = o.shape[0]
        u, up = _d(np.asarray(u, dtype=np.float64).reshape(n, max(max_depth, 0), 3))
        rgb = np.zeros((n, 3))
        rays = np.zeros(n, dtype=np.int32)
        self.lib.orc_trace(scene, n, op, dp, max_depth, up, rgb.ctypes.data_as(_dp), rays.ctypes.data_as(_ip))
        return dict(rgb=rgb, rays=rays)

    # -- render ------------------------------------------------------------------------------
    def render(self, scene, cam, W, H, spp_begin, spp_end, max_depth=50, mode=0, seed_a=1, seed_b=2, threads=1,
               shared_b=0, want_sumsq=False):
        cam, cp = _d(cam)
        s = np.zeros((H, W, 3))
        sq = np.zeros((H, W, 3)) if want_sumsq else None
        rays = C.c_uint64(0)
        self.lib.orc_render(scene, cp, W, H, spp_begin, spp_end, max_depth, mode, seed_a, seed_b, threads, shared_b,
                            s.ctypes.data_as(_dp), sq.ctypes.data_as(_dp) if want_sumsq else None, C.byref(rays))
        return s, sq, int(rays.value)

    def format_ppm(self, s, spp):
        s, sp = _d(s)
        H, W = s.shape[0], s.shape[1]
        cap = 32 + W * H * 12
        buf = C.create_string_buffer(cap)
        n = self.lib.orc_format_ppm(sp, W, H, spp, buf, cap)
        assert n > 0
        return buf.raw[:n]

"""ctypes binding of libpt_host.so: the host-side C++ that keeps the reference's interfaces
(config.txt reader, Camera / Hittable / Material class surface, scene generator, PPM writer).
Scene generation, camera set-up, config parsing and PPM formatting run on the host as in the
reference; every ray query of the class surface is answered by the gfx950 kernels.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libpt_host.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_fp = C.POINTER(C.c_float)

HOST_SIGNATURES = {
    "pth_last_error": (C.c_char_p, []),
    "pth_generate_scene": (C.c_int, [C.c_int, C.c_uint, C.c_uint, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                     _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp]),
    "pth_camera_setup": (None, [_dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _dp]),
    "pth_read_settings": (C.c_int, [C.c_char_p, _dp]),
    "pth_read_vector": (C.c_int, [C.c_char_p, _dp]),
    "pth_format_ppm": (C.c_long, [_fp, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_long]),
    "pth_world_is_hit": (C.c_int, [C.c_int, _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double,
                                   C.c_double, _dp, _dp, _dp, _ip, _ip]),
    "pth_sphere_is_hit": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _ip]),
    "pth_material_scatter": (C.c_int, [C.c_int, _dp, C.c_double, C.c_double, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp]),
    "pth_camera_ray": (C.c_int, [_dp, _dp, _dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                 C.c_double, C.c_double, C.c_double, _dp, _dp]),
    "pth_render_ppm": (C.c_long, [C.c_char_p, C.c_char_p, C.c_long, _dp]),
}

_lib = None
SCENE_KEYS = ("cx", "cy", "cz", "r", "ar", "ag", "ab", "fuzz", "ior")
FOCUS_DEFAULT = float(np.sqrt(77.0))   # Path Tracer.cpp:127: frozen at the default camera


def load_host_library():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HOST_LIB_PATH):
        raise RuntimeError("%s is missing: run `make` (or __graft_entry__.build()) first" % HOST_LIB_PATH)
    lib = C.CDLL(HOST_LIB_PATH)
    for name, (res, args) in HOST_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t=_dp):
    return a.ctypes.data_as(t)


def generate_scene(count=250, seed_a=1, seed_b=2, half_extent=10.0, diffuse_below=0.6, metal_below=0.9, use_grid=True):
    """Path Tracer.cpp:182-260: five fixed + up to `count` random spheres, in list order. use_grid: the overlap
    test goes through a uniform grid instead of the reference's scan of every object (same scene, bit for bit)."""
    lib = load_host_library()
    cap = count + 5
    f = {k: np.zeros(cap) for k in SCENE_KEYS}
    kind = np.zeros(cap, dtype=np.int32)
    n = lib.pth_generate_scene(count, seed_a, seed_b, half_extent, diffuse_below, metal_below, 1 if use_grid else 0, cap,
                               _p(f["cx"]), _p(f["cy"]), _p(f["cz"]), _p(f["r"]), _p(kind, _ip),
                               _p(f["ar"]), _p(f["ag"]), _p(f["ab"]), _p(f["fuzz"]), _p(f["ior"]))
    out = {k: v[:n].copy() for k, v in f.items()}
    out["kind"] = kind[:n].copy()
    return out


def camera_setup(pos=(8, 2, 3), look=(0, 0, 0), up=(0, 1, 0), aspect=16.0 / 9.0, focal=1.0, fov=60.0, aperture=0.1,
                 focus=FOCUS_DEFAULT):
    """Camera.cpp:7-32: position, lowerLeft, horizontal, vertical, u, v, lensRadius as 19 doubles."""
    lib = load_host_library()
    cam = np.zeros(19)
    lib.pth_camera_setup(_p(_d(pos)), _p(_d(look)), _p(_d(up)), aspect, focal, fov, aperture, focus, _p(cam))
    return cam


def read_settings(text):
    """Parses config text as the reference's main() does. Returns (dict, used_defaults)."""
    lib = load_host_library()
    out = np.zeros(26)
    used = lib.pth_read_settings(text.encode(), _p(out))
    s = dict(imageAspectRatio=out[0], imageWidth=int(out[1]), raysPerPixel=int(out[2]), materialMaxDepth=int(out[3]),
             cameraPosition=tuple(out[4:7]), cameraLookingAt=tuple(out[7:10]), cameraUpOrient=tuple(out[10:13]),
             focalLength=out[13], verticalFOV=out[14], apertureSize=out[15], focusDistance=out[16],
             imageHeight=int(out[17]),
             # optional keys of this build (absent: the reference's behaviour)
             numberOfSpheres=int(out[18]), sceneSeedA=int(out[19]), sceneSeedB=int(out[20]), renderSeed=int(out[21]),
             sceneHalfExtent=out[22], numberOfGPUs=int(out[23]), outputFormat="P6" if out[24] else "P3",
             closestHitSearch="grid" if out[25] else "scan")
    return s, bool(used)


def read_vector(text):
    lib = load_host_library()
    out = np.zeros(3)
    ok = lib.pth_read_vector(text.encode(), _p(out))
    return bool(ok), tuple(out)


def format_ppm(sum_rgb, spp):
    """The reference's ASCII P3 file (Path Tracer.cpp:62-74, :316-322) for sums over spp samples."""
    lib = load_host_library()
    a = np.ascontiguousarray(sum_rgb, dtype=np.float32)
    h, w = a.shape[0], a.shape[1]
    cap = 32 + w * h * 12
    buf = C.create_string_buffer(cap)
    n = lib.pth_format_ppm(_p(a, _fp), w, h, spp, buf, cap)
    if n <= 0:
        raise RuntimeError("pth_format_ppm failed")
    return buf.raw[:n]


def _scene_args(sc):
    keep = [_d(sc[k]) for k in ("cx", "cy", "cz", "r")]
    kind = np.ascontiguousarray(sc["kind"], dtype=np.int32)
    keep2 = [_d(sc[k]) for k in ("ar", "ag", "ab", "fuzz", "ior")]
    return keep, kind, keep2


def world_is_hit(sc, o, d, t_min=0.001, t_max=float("inf")):
    """HittableList::isHit through the host classes (device-backed). Returns None on a miss."""
    lib = load_host_library()
    keep, kind, keep2 = _scene_args(sc)
    t = C.c_double(0)
    p = np.zeros(3)
    n = np.zeros(3)
    front = C.c_int(0)
    mk = C.c_int(0)
    rc = lib.pth_world_is_hit(len(kind), *[_p(k) for k in keep], _p(kind, _ip), *[_p(k) for k in keep2], _p(_d(o)),
                              _p(_d(d)), t_min, t_max, C.byref(t), _p(p), _p(n), C.byref(front), C.byref(mk))
    if rc < 0:
        raise RuntimeError(lib.pth_last_error().decode())
    if rc == 0:
        return None
    return dict(t=t.value, p=p, n=n, front=front.value, kind=mk.value)


def sphere_is_hit(c, r, o, d, t_min=0.001, t_max=float("inf")):
    lib = load_host_library()
    t = C.c_double(0)
    p = np.zeros(3)
    n = np.zeros(3)
    front = C.c_int(0)
    rc = lib.pth_sphere_is_hit(_p(_d(c)), r, _p(_d(o)), _p(_d(d)), t_min, t_max, C.byref(t), _p(p), _p(n), C.byref(front))
    if rc < 0:
        raise RuntimeError(lib.pth_last_error().decode())
    if rc == 0:
        return None
    return dict(t=t.value, p=p, n=n, front=front.value)


def material_scatter(kind, albedo, fuzz, ior, d_in, p, nrm, front, u):
    lib = load_host_library()
    d_out = np.zeros(3)
    atten = np.zeros(3)
    rc = lib.pth_material_scatter(kind, _p(_d(albedo)), fuzz, ior, _p(_d(d_in)), _p(_d(p)), _p(_d(nrm)), int(front),
                                  _p(_d(u)), _p(d_out), _p(atten))
    if rc < 0:
        raise RuntimeError(lib.pth_last_error().decode())
    return dict(ok=rc, d=d_out, atten=atten)


def camera_ray(x, y, diskx, disky, pos=(8, 2, 3), look=(0, 0, 0), up=(0, 1, 0), aspect=16.0 / 9.0, focal=1.0, fov=60.0,
               aperture=0.1, focus=FOCUS_DEFAULT):
    lib = load_host_library()
    o = np.zeros(3)
    d = np.zeros(3)
    rc = lib.pth_camera_ray(_p(_d(pos)), _p(_d(look)), _p(_d(up)), aspect, focal, fov, aperture, focus, x, y, diskx, disky,
                            _p(o), _p(d))
    if rc < 0:
        raise RuntimeError(lib.pth_last_error().decode())
    return o, d


def render_ppm(config_text, cap=1 << 26):
    """config text -> scene -> device render -> P3 bytes, all through the host classes."""
    lib = load_host_library()
    buf = C.create_string_buffer(cap)
    rate = C.c_double(0)
    n = lib.pth_render_ppm(config_text.encode(), buf, cap, C.byref(rate))
    if n <= 0:
        raise RuntimeError("pth_render_ppm failed: " + lib.pth_last_error().decode())
    return buf.raw[:n], rate.value

"""Python binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY — see oracle/oracle.h.  Importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never from pooraytracer_amd/.  PARITY UNPINNED (oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

D3 = C.c_double * 3


class OrcMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("texture", C.c_int32), ("kd", D3), ("ks", D3), ("ns", C.c_double),
                ("emission", D3), ("eta", D3), ("k", D3), ("alpha_x", C.c_double), ("alpha_y", C.c_double)]


class OrcTexture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("reserved", C.c_int32),
                ("data", C.c_void_p)]


class OrcSceneDesc(C.Structure):
    _fields_ = [("n_tris", C.c_uint64), ("vertices", C.c_void_p), ("normals", C.c_void_p), ("texcoords", C.c_void_p),
                ("n_meshes", C.c_uint32), ("n_materials", C.c_uint32), ("mesh_first_tri", C.c_void_p),
                ("mesh_material", C.c_void_p), ("materials", C.c_void_p), ("n_textures", C.c_uint32),
                ("reserved", C.c_uint32), ("textures", C.c_void_p), ("light_meshes", C.c_void_p),
                ("n_light_meshes", C.c_uint32), ("reserved2", C.c_uint32)]


class OrcCamera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fovy", C.c_double), ("eye", D3), ("look_at", D3),
                ("up", D3)]


class OrcRenderParams(C.Structure):
    _fields_ = [("spp", C.c_int32), ("max_depth", C.c_int32), ("russian_roulette", C.c_double),
                ("sample_lights", C.c_int32), ("precision", C.c_int32), ("background", D3), ("seed", C.c_uint64),
                ("tile_size", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32), ("sample_chunks", C.c_int32),
                ("pixel_jitter", C.c_int32), ("reserved", C.c_int32)]


class OrcCounters(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("hit_calls", C.c_uint64),
                ("samples", C.c_uint64)]


RAY_DTYPE = np.dtype([("o", "<f8", 3), ("tmin", "<f8"), ("d", "<f8", 3), ("tmax", "<f8")])
HIT_DTYPE = np.dtype([("t", "<f8"), ("alpha", "<f8"), ("beta", "<f8"), ("prim", "<i4"), ("front", "<i4")])
LIGHT_SAMPLE_DTYPE = np.dtype(
    [("position", "<f8", 3), ("normal", "<f8", 3), ("pdf", "<f8"), ("prim", "<i4"), ("front", "<i4")])


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc only; runs anywhere)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
            os.path.getmtime(os.path.join(_HERE, f)) for f in ("pt_oracle.cpp", "oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.c_void_p]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_light_count.restype = C.c_uint64
        L.orc_light_count.argtypes = [C.c_void_p]
        L.orc_light_order.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_sample_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p]
        L.orc_rng_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p]
        L.orc_render_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_render_samples_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32,
                                               C.c_void_p, C.c_void_p]
        L.orc_camera_rays.argtypes = [C.c_void_p, C.c_void_p]
        vp = C.c_void_p
        L.orc_material_eval.argtypes = [vp, C.c_int32, C.c_size_t, vp, vp, vp, C.c_uint64, vp]
        L.orc_material_scatter.argtypes = [vp, C.c_int32, C.c_size_t, vp, vp, vp, vp, C.c_uint64, vp, vp, vp]
        L.orc_texture_value.argtypes = [vp, C.c_int32, C.c_size_t, vp, vp]
        L.orc_cooktorrance_terms.argtypes = [vp, C.c_int32, C.c_size_t, vp, vp, vp, vp]
        L.orc_frame.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, vp]
        _lib = L
    return _lib


def _marshal(scene):
    # same layout as the product ABI; re-marshalled here so the oracle has no product dependency
    from pooraytracer_amd import _abi  # data-marshalling helper only (no compute)
    return _abi.marshal_scene(scene, OrcSceneDesc, OrcMaterial, OrcTexture)


def _cam(cam):
    c = OrcCamera()
    c.width, c.height, c.fovy = cam.width, cam.height, cam.fovy
    c.eye, c.look_at, c.up = D3(*cam.eye), D3(*cam.look_at), D3(*cam.up)
    return c


def _params(spp=1, max_depth=10, rr=0.8, sample_lights=True, background=(0.0, 0.0, 0.0), seed=1, pixel_jitter=False):
    p = OrcRenderParams()
    p.spp, p.max_depth, p.russian_roulette = spp, max_depth, rr
    p.sample_lights = int(bool(sample_lights))
    p.background = D3(*background)
    p.seed, p.tile_size, p.rank, p.nranks = seed, 32, 0, 1
    p.pixel_jitter = int(bool(pixel_jitter))
    return p


class Oracle:
    """CPU oracle scene: restated reference object graph (world + lights two-level BVH)."""

    def __init__(self, scene):
        self.scene = scene
        desc, self._keep = _marshal(scene)
        self._h = lib().orc_scene_create(C.byref(desc))

    def close(self):
        if self._h:
            lib().orc_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def light_order(self):
        n = lib().orc_light_count(self._h)
        out = np.zeros(n, dtype=np.int32)
        if n:
            lib().orc_light_order(self._h, out.ctypes.data)
        return out

    def trace_closest(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        lib().orc_trace_closest(self._h, rays.ctypes.data, rays.shape[0], hits.ctypes.data)
        return hits

    def sample_lights(self, origins, seed=1):
        origins = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(origins.shape[0], dtype=LIGHT_SAMPLE_DTYPE)
        lib().orc_sample_lights(self._h, origins.ctypes.data, origins.shape[0], seed, out.ctypes.data)
        return out

    # ---- material / texture hooks (known-answer tests); same signatures as pooraytracer_amd.api.Scene's
    def material_eval(self, material, wi, wo, uv=None, seed=1):
        wi, wo = _f64(wi, 3), _f64(wo, 3)
        uv = None if uv is None else _f64(uv, 2)
        out = np.zeros_like(wi)
        lib().orc_material_eval(self._h, material, wi.shape[0], wi.ctypes.data, wo.ctypes.data,
                                None if uv is None else uv.ctypes.data, seed, out.ctypes.data)
        return out

    def material_scatter(self, material, rd, normal=(0, 0, 1), tangent=(1, 0, 0), uv=None, seed=1):
        rd = _f64(rd, 3)
        uv = None if uv is None else _f64(uv, 2)
        nrm, tan = _f64(normal, 3), _f64(tangent, 3)
        wi, att, ok = np.zeros_like(rd), np.zeros_like(rd), np.zeros(rd.shape[0], dtype=np.int32)
        lib().orc_material_scatter(self._h, material, rd.shape[0], rd.ctypes.data, nrm.ctypes.data, tan.ctypes.data,
                                   None if uv is None else uv.ctypes.data, seed, wi.ctypes.data, att.ctypes.data, ok.ctypes.data)
        return wi, att, ok.astype(bool)

    def texture_value(self, texture, uv):
        uv = _f64(uv, 2)
        out = np.zeros((uv.shape[0], 3))
        lib().orc_texture_value(self._h, texture, uv.shape[0], uv.ctypes.data, out.ctypes.data)
        return out

    def cooktorrance_terms(self, material, w, wm):
        w, wm = _f64(w, 3), _f64(wm, 3)
        out, fr = np.zeros((w.shape[0], 4)), np.zeros((w.shape[0], 3))
        lib().orc_cooktorrance_terms(self._h, material, w.shape[0], w.ctypes.data, wm.ctypes.data, out.ctypes.data, fr.ctypes.data)
        return {"D": out[:, 0], "Lambda": out[:, 1], "G1": out[:, 2], "Dv": out[:, 3], "F": fr}

    def render(self, camera=None, nthreads=1, reuse_peek=True, rows=None, **kw):
        cam = camera or self.scene.camera
        c, p = _cam(cam), _params(**kw)
        rgb = np.zeros((cam.height, cam.width, 3), dtype=np.float64)
        cnt = OrcCounters()
        y0, y1 = rows if rows else (0, cam.height)
        lib().orc_render(self._h, C.byref(c), C.byref(p), rgb.ctypes.data, y0, y1, nthreads, int(reuse_peek),
                         C.byref(cnt))
        counters = {f: getattr(cnt, f) for f, _ in OrcCounters._fields_}
        return rgb, counters

    def render_samples(self, pixels_xy, camera=None, sample_begin=0, sample_count=None, trace=False, **kw):
        """RayColor per (pixel, sample): (n, count, 3); with trace=True also the path signatures (n, count, TRACE_WORDS)."""
        cam = camera or self.scene.camera
        c, p = _cam(cam), _params(**kw)
        count = p.spp if sample_count is None else int(sample_count)
        px = np.ascontiguousarray(pixels_xy, dtype=np.int32).reshape(-1, 2)
        out = np.zeros((px.shape[0], count, 3), dtype=np.float64)
        tr = np.zeros((px.shape[0], count, TRACE_WORDS), dtype=np.int32) if trace else None
        lib().orc_render_samples_trace(self._h, C.byref(c), C.byref(p), px.ctypes.data, px.shape[0], int(sample_begin), count,
                                       out.ctypes.data, tr.ctypes.data if trace else None)
        return (out, tr) if trace else out


TRACE_WORDS = 64


def _f64(a, k):
    return np.ascontiguousarray(a, dtype=np.float64).reshape(-1, k)


def frame(normal, tangent, v, to_local):
    """Material::WorldToLocal (to_local) / LocalToWorld in the frame (normal, tangent)."""
    v = _f64(v, 3)
    out = np.zeros_like(v)
    lib().orc_frame(_f64(normal, 3).ctypes.data, _f64(tangent, 3).ctypes.data, v.shape[0], v.ctypes.data, int(bool(to_local)), out.ctypes.data)
    return out


def rng_stream(seed, pixel, sample, n):
    out = np.zeros(n, dtype=np.float64)
    lib().orc_rng_stream(seed, pixel, sample, n, out.ctypes.data)
    return out


def camera_rays(cam):
    c = _cam(cam)
    out = np.zeros((cam.height, cam.width, 6), dtype=np.float64)
    lib().orc_camera_rays(C.byref(c), out.ctypes.data)
    return out

"""ctypes mirror of include/prt.h (the C ABI of libprt_hip.so).

Plumbing only: plain structs, pointers and sizes.  Field order/types must match prt.h exactly;
tests/test_abi.py checks sizes and that every declared symbol is exported.
"""
import ctypes as C

PRT_ABI_VERSION = 5
TRACE_WORDS, TRACE_VERTS = 64, 31           # prt.h PRT_TRACE_*
TRACE_NEE, TRACE_VISIBLE, TRACE_ROULETTE, TRACE_SCATTER = 1, 2, 4, 8
PRECISION_F64, PRECISION_F32 = 0, 1

PRT_OK = 0
PRT_E_INVALID = -1
PRT_E_NO_DEVICE = -2
PRT_E_HIP = -3
PRT_E_OOM = -4
PRT_E_LIMIT = -5

MAT_LAMBERTIAN = 0
MAT_PHONG = 1
MAT_MIRROR = 2
MAT_COOKTORRANCE = 3
MAT_DIFFUSE_LIGHT = 4
MAT_DEBUG = 5
MAT_EMPTY = 6

D3 = C.c_double * 3


class PrtMaterial(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("texture", C.c_int32),
        ("kd", D3),
        ("ks", D3),
        ("ns", C.c_double),
        ("emission", D3),
        ("eta", D3),
        ("k", D3),
        ("alpha_x", C.c_double),
        ("alpha_y", C.c_double),
    ]


class PrtTexture(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("channels", C.c_int32),
        ("reserved", C.c_int32),
        ("data", C.c_void_p),
    ]


class PrtSceneDesc(C.Structure):
    _fields_ = [
        ("n_tris", C.c_uint64),
        ("vertices", C.c_void_p),
        ("normals", C.c_void_p),
        ("texcoords", C.c_void_p),
        ("n_meshes", C.c_uint32),
        ("n_materials", C.c_uint32),
        ("mesh_first_tri", C.c_void_p),
        ("mesh_material", C.c_void_p),
        ("materials", C.c_void_p),
        ("n_textures", C.c_uint32),
        ("flags", C.c_uint32),
        ("textures", C.c_void_p),
        ("light_meshes", C.c_void_p),
        ("n_light_meshes", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


PRT_SCENE_DEVICE_BVH = 1


class PrtBvhInfo(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_uint64),
        ("depth", C.c_uint32),
        ("built_on_device", C.c_uint32),
        ("build_ms", C.c_double),
        ("sort_ms", C.c_double),
        ("tree_ms", C.c_double),
        ("split_ms", C.c_double),
        ("node_bytes", C.c_uint32),
        ("width", C.c_uint32),
        ("tri_bytes", C.c_uint32),
        ("tri_stride", C.c_uint32),
        ("texture_bytes", C.c_uint64),
        ("texture_footprint_bytes", C.c_uint64),
        ("texture_layouts", C.c_uint32),
        ("render_blocks_per_cu", C.c_uint32),
        ("render_blocks_wanted", C.c_uint32),
        ("lds_materials", C.c_uint32),
        ("lds_light_nodes", C.c_uint32),
        ("lds_light_tris", C.c_uint32),
        ("stack_need", C.c_uint32),
        ("reserved_", C.c_uint32),
    ]


class PrtCamera(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("fovy", C.c_double),
        ("eye", D3),
        ("look_at", D3),
        ("up", D3),
    ]


class PrtRenderParams(C.Structure):
    _fields_ = [
        ("spp", C.c_int32),
        ("max_depth", C.c_int32),
        ("russian_roulette", C.c_double),
        ("sample_lights", C.c_int32),
        ("precision", C.c_int32),
        ("background", D3),
        ("seed", C.c_uint64),
        ("tile_size", C.c_int32),
        ("rank", C.c_int32),
        ("nranks", C.c_int32),
        ("sample_chunks", C.c_int32),
        ("pixel_jitter", C.c_int32),
        ("reserved", C.c_int32),
    ]


class PrtRay(C.Structure):
    _fields_ = [("o", D3), ("tmin", C.c_double), ("d", D3), ("tmax", C.c_double)]


class PrtHit(C.Structure):
    _fields_ = [
        ("t", C.c_double),
        ("alpha", C.c_double),
        ("beta", C.c_double),
        ("prim", C.c_int32),
        ("front", C.c_int32),
    ]


class PrtLightSample(C.Structure):
    _fields_ = [
        ("position", D3),
        ("normal", D3),
        ("pdf", C.c_double),
        ("prim", C.c_int32),
        ("front", C.c_int32),
    ]


class PrtCounters(C.Structure):
    _fields_ = [
        ("rays_closest", C.c_uint64),
        ("rays_shadow", C.c_uint64),
        ("node_fetches", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("samples", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("bvh_nodes", C.c_uint64),
        ("bvh_depth", C.c_uint64),
        ("inner_rounds", C.c_uint64),
        ("leaf_rounds", C.c_uint64),
        ("refills", C.c_uint64),
        ("tri_full", C.c_uint64),
    ]


# numpy structured dtypes with the same layout (for zero-copy batches)
import numpy as np  # noqa: E402

RAY_DTYPE = np.dtype([("o", "<f8", 3), ("tmin", "<f8"), ("d", "<f8", 3), ("tmax", "<f8")])
HIT_DTYPE = np.dtype([("t", "<f8"), ("alpha", "<f8"), ("beta", "<f8"), ("prim", "<i4"), ("front", "<i4")])
LIGHT_SAMPLE_DTYPE = np.dtype(
    [("position", "<f8", 3), ("normal", "<f8", 3), ("pdf", "<f8"), ("prim", "<i4"), ("front", "<i4")]
)
assert RAY_DTYPE.itemsize == C.sizeof(PrtRay) == 64
assert HIT_DTYPE.itemsize == C.sizeof(PrtHit) == 32
assert LIGHT_SAMPLE_DTYPE.itemsize == C.sizeof(PrtLightSample) == 64

# every symbol include/prt.h declares
EXPORTS = [
    "prt_abi_version",
    "prt_dev_hooks",
    "prt_shutdown",
    "prt_render_samples",
    "prt_render_multi",
    "prt_last_error",
    "prt_device_count",
    "prt_scene_create",
    "prt_scene_destroy",
    "prt_scene_upload",
    "prt_scene_update_vertices",
    "prt_scene_bvh_info",
    "prt_scene_light_count",
    "prt_scene_light_order",
    "prt_trace_closest",
    "prt_trace_closest_device",
    "prt_trace_closest_device_prec",
    "prt_trace_closest_sorted_device",
    "prt_sample_lights",
    "prt_render",
    "prt_render_device",
    "prt_get_counters",
    "prt_tonemap_srgb8",
    "prt_material_eval",
    "prt_material_scatter",
    "prt_texture_value",
]


def marshal_scene(scene, desc_cls=PrtSceneDesc, mat_cls=PrtMaterial, tex_cls=PrtTexture):
    """Build a (desc, keepalive) pair from a scenes.SceneData.  `keepalive` owns the buffers."""
    keep = []
    v = np.ascontiguousarray(scene.vertices, dtype=np.float64)
    n = None if scene.normals is None else np.ascontiguousarray(scene.normals, dtype=np.float64)
    t = None if scene.texcoords is None else np.ascontiguousarray(scene.texcoords, dtype=np.float64)
    first = np.ascontiguousarray(scene.mesh_first_tri, dtype=np.uint64)
    mm = np.ascontiguousarray(scene.mesh_material, dtype=np.int32)
    keep += [v, n, t, first, mm]
    mats = (mat_cls * max(1, len(scene.materials)))()
    for i, m in enumerate(scene.materials):
        mats[i].type = m.type
        mats[i].texture = m.texture
        mats[i].kd = D3(*m.kd)
        mats[i].ks = D3(*m.ks)
        mats[i].ns = m.ns
        mats[i].emission = D3(*m.emission)
        mats[i].eta = D3(*m.eta)
        mats[i].k = D3(*m.k)
        mats[i].alpha_x = m.alpha_x
        mats[i].alpha_y = m.alpha_y
    texs = (tex_cls * max(1, len(scene.textures)))()
    for i, tx in enumerate(scene.textures):
        arr = np.ascontiguousarray(tx, dtype=np.uint8)
        keep.append(arr)
        texs[i].height, texs[i].width = arr.shape[0], arr.shape[1]
        texs[i].channels = arr.shape[2] if arr.ndim == 3 else 1
        texs[i].data = arr.ctypes.data
    keep += [mats, texs]
    d = desc_cls()
    d.n_tris = v.shape[0]
    d.vertices = v.ctypes.data
    d.normals = None if n is None else n.ctypes.data
    d.texcoords = None if t is None else t.ctypes.data
    d.n_meshes = len(mm)
    d.n_materials = len(scene.materials)
    d.mesh_first_tri = first.ctypes.data
    d.mesh_material = mm.ctypes.data
    d.materials = C.addressof(mats)
    d.n_textures = len(scene.textures)
    d.textures = C.addressof(texs)
    lm = getattr(scene, "light_meshes", None)  # the `lights` list of Camera::Render as mesh indices; None = main.cpp's
    if lm is not None:
        arr = np.ascontiguousarray(lm, dtype=np.int32).reshape(-1)
        buf = (C.c_int32 * max(1, arr.size))(*arr.tolist())
        keep.append(buf)
        d.light_meshes = C.addressof(buf)
        d.n_light_meshes = arr.size
    return d, keep


def make_camera(cam, cls=PrtCamera):
    c = cls()
    c.width, c.height, c.fovy = cam.width, cam.height, cam.fovy
    c.eye, c.look_at, c.up = D3(*cam.eye), D3(*cam.look_at), D3(*cam.up)
    return c


def make_params(cls=PrtRenderParams, spp=1, max_depth=10, rr=0.8, sample_lights=True, background=(0.0, 0.0, 0.0),
                seed=1, tile_size=32, rank=0, nranks=1, sample_chunks=0, precision=0,
                pixel_jitter=False):
    p = cls()
    p.spp, p.max_depth, p.russian_roulette = spp, max_depth, rr
    p.sample_lights, p.precision = int(bool(sample_lights)), precision
    p.background = D3(*background)
    p.seed, p.tile_size, p.rank, p.nranks, p.sample_chunks = seed, tile_size, rank, nranks, sample_chunks
    p.pixel_jitter = int(bool(pixel_jitter))
    return p

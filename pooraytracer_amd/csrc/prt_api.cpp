// prt_api.cpp — the C ABI of libprt_hip.so (include/prt.h).  Host orchestration only: scene
// preparation on create, SoA upload, kernel launches, counters.  There is no CPU compute path:
// every compute entry point needs a HIP device and fails with PRT_E_NO_DEVICE / PRT_E_HIP otherwise.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "prt_host.h"

namespace prt {
int render_blocks_per_cu(bool count, int feat, size_t table_bytes, int stack_depth, bool pad, bool extra);
int render_permutation(int feat);
int render_lds_budget(int feat, int stack_depth);
size_t render_table_bytes(int light_lds, int mat_lds, int ltri_lds);
void launch_trace(const DScene& S, const PrtRay* d_rays, size_t n, PrtHit* d_hits, DCounters* d_ctr, bool count, int n_cu,
                  hipStream_t st, const uint32_t* d_perm = nullptr);
// K4 (ray_sort.hip): a permutation of a ray batch in which consecutive rays start close together
size_t ray_sort_scratch_bytes(size_t n, std::string* err);
const uint32_t* ray_sort(const PrtRay* d_rays, size_t n, const float grid_origin[3], const float grid_step[3], void* scratch,
                         size_t scratch_bytes, hipStream_t st, std::string* err);
void launch_render(const DScene& S, const DCamera& C, const DRenderParams& P, double* d_partial, DCounters* d_ctr,
                   bool count, int feat, unsigned grid, hipStream_t st);
void launch_finalize(const DCamera& C, const DRenderParams& P, const double* d_partial, double* d64, float* d32,
                     hipStream_t st);
void launch_sample_lights(const DScene& S, const double* d_origins, size_t n, uint64_t seed, PrtLightSample* d_out,
                          hipStream_t st);
void launch_tonemap(const float* d_in, size_t n, uint8_t* d_out, hipStream_t st);
void launch_add_f32(float* dst, const float* src, size_t n, hipStream_t st);
void launch_material_eval(const DScene& S, int material, const double* wi, const double* wo, const double* uv, size_t n,
                          uint64_t seed, double* out, hipStream_t st);
void launch_material_scatter(const DScene& S, int material, const double* rd, const double* normal, const double* tangent,
                             const double* uv, size_t n, uint64_t seed, double* wi_out, double* att_out, int32_t* ok_out,
                             hipStream_t st);
void launch_texture_value(const DScene& S, int texture, const double* uv, size_t n, double* out, hipStream_t st);
void launch_gather_tris(const DTri* tri_in, const DTriShade* shade_in, const uint32_t* order, uint32_t n, void* tri_out,
                        uint32_t tri_out_stride, DTriShade* shade_out, hipStream_t st);
} // namespace prt

// fp32 fast mode (prt_kernels_f32.hip): K1 and K3 on float records derived from the resident fp64 ones
namespace prt32 {
typedef DSceneT<float> Scene32;
int render_blocks_per_cu(bool count, int feat, size_t table_bytes, int stack_depth, bool pad, bool extra);
int render_lds_budget(int feat, int stack_depth);
size_t render_table_bytes(int light_lds, int mat_lds, int ltri_lds);
void launch_trace(const Scene32& S, const PrtRay* d_rays, size_t n, PrtHit* d_hits, DCounters* d_ctr, bool count, int n_cu,
                  hipStream_t st, const uint32_t* d_perm = nullptr);
void launch_render(const Scene32& S, const DCameraT<float>& C, const DRenderParamsT<float>& P, double* d_partial,
                   DCounters* d_ctr, bool count, int feat, unsigned grid, hipStream_t st);
void launch_convert_tris(const void* in, uint32_t in_stride, uint32_t n, void* out, uint32_t out_stride, hipStream_t st);
void launch_convert_shade(const DTriShadeT<double>* in, uint32_t n, DTriShadeT<float>* out, hipStream_t st);
void launch_convert_reals(const double* in, size_t n, float* out, hipStream_t st);
} // namespace prt32

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
// Developer / test hooks are environment variables (PRT_TUNE_*: scheduling thresholds, table sizes, layouts for A/B runs;
// PRT_TEST_*: failure injection; PRT_VALIDATE_BVH).  The shipped libprt_hip.so does NOT read them: only a build with
// -DPRT_DEV_HOOKS=1 does (pooraytracer_amd/build.py builds that one as libprt_hip_dev.so for the tests and sweep tools
// that need it), so no environment variable can change what the production library schedules or make it fail.
#ifndef PRT_DEV_HOOKS
#define PRT_DEV_HOOKS 0
#endif
#if PRT_DEV_HOOKS
const char* dev_env(const char* name) { return std::getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; } // (not constexpr: call sites pass the result on to atoi)
#endif
} // namespace

extern "C" int prt_dev_hooks(void) { return PRT_DEV_HOOKS; }

#define PRT_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? PRT_E_OOM : PRT_E_HIP,                              \
                        std::string(#call) + ": " + hipGetErrorString(e_));                             \
    } while (0)

struct PrtScene {
    // host side
    std::vector<prt::HostTri> tris;
    std::vector<DMaterial> mats;
    std::vector<DTexture> texs;
    std::vector<double> texels_lin; // GetPixel() of every texel (Texture.cpp:50-65)
    size_t n_texel_reals = 0;       // reals of the texel array resident on the device (16 per texel as footprints, 3 as plain texels)
    size_t tex_footprint_bytes = 0; // ... of which footprint records (fp64 bytes)
    uint32_t tex_layouts = 0;       // bit 0: some texture is stored as footprints, bit 1: some as plain texels
    prt::LightTree lights;
    prt::BuiltBVH bvh;
    std::vector<uint64_t> mesh_first; // mesh structure, kept for prt_scene_update_vertices
    std::vector<int32_t> mesh_mat;
    std::vector<int32_t> light_meshes; // PrtSceneDesc.light_meshes when given
    bool explicit_lights = false;
    bool device_bvh = false; // PRT_SCENE_DEVICE_BVH: the tree is built in prt_scene_upload, on the GPU
    PrtBvhInfo bvh_info{};
    // device side
    int device = -1;
    int n_cu = 0;
    int blocks_per_cu[2] = {0, 0};
    int blocks_wanted = 0; // what the production kernel's register allocation allows (occupancy without LDS tables)
    int ltri_lds = 0;
    int light_lds = 0, mat_lds = 0; // light-tree nodes / materials staged in LDS by K3 (both 0 = the kernels without LDS tables)
    int feat = 0; // material features of the scene (1 textures, 2 Phong, 4 CookTorrance) -> K3 permutation
    DScene d{};
    // fp32 fast mode: float copies of the tables, made on the first PRT_PRECISION_F32 call (ensure_f32)
    DSceneT<float> d32{};
    bool f32_ready = false;
    int blocks_per_cu32[2] = {0, 0};
    int ltri_lds32 = 0, light_lds32 = 0, mat_lds32 = 0; // table sizes of the fp32 kernels (their own LDS budget)
    int stack_depth32 = PRT_STACK_DEPTH;                // LDS stack entries per lane of the fp32 render kernels
    int stack_need = PRT_STACK_DEPTH;                   // stack entries a traversal of the resident tree can need at most
    int stack_depth = PRT_STACK_DEPTH;                  // ... and what the fp64 render kernels get: that, rounded up to a multiple of 4
    const DNode* d_nodes_shallow = nullptr;             // the same binary tree collapsed for PRT_STACK_SHALLOW entries (host build, deep trees), or null
    uint32_t n_nodes_shallow = 0;
    DScene d_k3{};                                      // the scene as K3 sees it: `d`, on the shallow tree when that buys LDS the kernel needs
    std::vector<void*> allocs;
    // Per-call device state, double-buffered: consecutive calls alternate slots, so a caller that
    // alternates two streams (and two framebuffers) can have frame k+1 filling the GPU while the last
    // long paths of frame k drain — the two launches never share counters, partial sums or events.
    struct CallSlot {
        DCounters* d_ctr = nullptr;
        double* d_partial = nullptr;
        size_t partial_cap = 0;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        hipEvent_t done = nullptr; // recorded behind the last kernel of the call that used this slot
        bool timed = false;
        bool counted = false;
        uint64_t samples = 0; // camera samples of the call (render: owned pixels inside the image x spp; the kernel does not count them)
    };
    // prt_trace_closest_sorted_device: K4's keys, values and the permutation — one scratch area per scene, kept between
    // calls; a call on another stream first waits for the previous sorted call's end (sort_done)
    void* d_sort = nullptr;
    size_t sort_cap = 0;
    hipEvent_t sort_done = nullptr;
    CallSlot slots[2];
    int cur = 0; // slot of the most recent call (prt_get_counters reads it)
    // Next slot for an asynchronous call on `st`.  A slot may still be in use by a call issued two calls ago on
    // another stream (three streams, or a trace call between two pipelined renders): its counters and partial
    // sums must not be reset under a running kernel, so the new call's stream first waits for that call's end.
    CallSlot* next_slot(hipStream_t st, hipError_t* err) {
        cur ^= 1;
        CallSlot& q = slots[cur];
        *err = q.timed ? hipStreamWaitEvent(st, q.done, 0) : hipSuccess;
        return &q;
    }
    PrtCounters last{};
    float* multi_fb = nullptr; // prt_render_multi: this device's full-size fp32 framebuffer (kept between frames)
    size_t multi_fb_cap = 0;

    int fail_upload_at = -1, n_uploads = 0; // test hook (PRT_TEST_FAIL_UPLOAD=k): the k-th table upload reports out-of-memory
    template <typename T>
    int up(const std::vector<T>& v, const T** out) {
        if (n_uploads++ == fail_upload_at) return fail(PRT_E_OOM, "prt_scene_upload: injected allocation failure (PRT_TEST_FAIL_UPLOAD)");
        void* p = nullptr;
        size_t bytes = std::max<size_t>(v.size() * sizeof(T), 256);
        PRT_HIP(hipMalloc(&p, bytes));
        allocs.push_back(p);
        if (!v.empty()) PRT_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        *out = reinterpret_cast<const T*>(p);
        return PRT_OK;
    }
    void release() {
        if (device >= 0) (void)hipSetDevice(device);
        for (void* p : allocs) (void)hipFree(p);
        allocs.clear();
        if (multi_fb) (void)hipFree(multi_fb);
        multi_fb = nullptr;
        multi_fb_cap = 0;
        if (d_sort) (void)hipFree(d_sort);
        d_sort = nullptr;
        sort_cap = 0;
        if (sort_done) (void)hipEventDestroy(sort_done);
        sort_done = nullptr;
        for (CallSlot& q : slots) {
            if (q.d_ctr) (void)hipFree(q.d_ctr);
            if (q.d_partial) (void)hipFree(q.d_partial);
            if (q.ev0) (void)hipEventDestroy(q.ev0);
            if (q.ev1) (void)hipEventDestroy(q.ev1);
            if (q.done) (void)hipEventDestroy(q.done);
            q = CallSlot();
        }
        device = -1;
        f32_ready = false;
    }
};

template <typename T, typename U>
static void conv_arr(T* o, const U* a, int n) {
    for (int i = 0; i < n; ++i) o[i] = (T)a[i];
}

extern "C" {

int prt_abi_version(void) { return PRT_ABI_VERSION; }
const char* prt_last_error(void) { return g_err.c_str(); }

int prt_device_count(int* n) {
    if (!n) return fail(PRT_E_INVALID, "prt_device_count: null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *n = 0;
        return fail(PRT_E_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *n = c;
    return PRT_OK;
}

int prt_scene_create(const PrtSceneDesc* desc, PrtScene** out) {
    if (!desc || !out) return fail(PRT_E_INVALID, "prt_scene_create: null argument");
    *out = nullptr;
    if (desc->n_tris && !desc->vertices) return fail(PRT_E_INVALID, "prt_scene_create: vertices is null");
    if (desc->n_meshes && (!desc->mesh_first_tri || !desc->mesh_material))
        return fail(PRT_E_INVALID, "prt_scene_create: mesh arrays are null");
    if (desc->n_materials && !desc->materials) return fail(PRT_E_INVALID, "prt_scene_create: materials is null");
    if (desc->n_textures && !desc->textures) return fail(PRT_E_INVALID, "prt_scene_create: textures is null");
    if (desc->n_meshes) {
        if (desc->mesh_first_tri[0] != 0 || desc->mesh_first_tri[desc->n_meshes] != desc->n_tris)
            return fail(PRT_E_INVALID, "prt_scene_create: mesh_first_tri must start at 0 and end at n_tris");
        for (uint32_t m = 0; m < desc->n_meshes; ++m) {
            if (desc->mesh_first_tri[m] > desc->mesh_first_tri[m + 1])
                return fail(PRT_E_INVALID, "prt_scene_create: mesh_first_tri must be ascending");
            if (desc->mesh_material[m] < 0 || (uint32_t)desc->mesh_material[m] >= desc->n_materials)
                return fail(PRT_E_INVALID, "prt_scene_create: mesh_material out of range");
        }
    } else if (desc->n_tris) {
        return fail(PRT_E_INVALID, "prt_scene_create: triangles without meshes");
    }
    if (desc->n_light_meshes && !desc->light_meshes) return fail(PRT_E_INVALID, "prt_scene_create: light_meshes is null");
    for (uint32_t i = 0; i < (desc->light_meshes ? desc->n_light_meshes : 0u); ++i)
        if (desc->light_meshes[i] < 0 || (uint32_t)desc->light_meshes[i] >= desc->n_meshes)
            return fail(PRT_E_INVALID, "prt_scene_create: light_meshes entry out of range");
    for (uint32_t i = 0; i < desc->n_materials; ++i) {
        const PrtMaterial& m = desc->materials[i];
        if (m.type < PRT_MAT_LAMBERTIAN || m.type > PRT_MAT_EMPTY)
            return fail(PRT_E_INVALID, "prt_scene_create: unknown material type");
        if (m.texture >= (int32_t)desc->n_textures) return fail(PRT_E_INVALID, "prt_scene_create: texture index out of range");
    }
    // NaN / infinite coordinates have no place in a BVH (the builders' orderings would be inconsistent)
    for (uint64_t i = 0; i < desc->n_tris * 9; ++i)
        if (!(std::fabs(desc->vertices[i]) <= 1e18)) // also catches NaN; the fp32 box tests need headroom below FLT_MAX
            return fail(PRT_E_INVALID, "prt_scene_create: vertex coordinate is not finite (or beyond 1e18)");
    PrtScene* s = new (std::nothrow) PrtScene();
    if (!s) return fail(PRT_E_OOM, "prt_scene_create: out of host memory");
    try {
        prt::setup_triangles(*desc, s->tris);
        s->mesh_first.assign(desc->mesh_first_tri, desc->mesh_first_tri + (desc->n_meshes ? desc->n_meshes + 1 : 0));
        s->mesh_mat.assign(desc->mesh_material, desc->mesh_material + desc->n_meshes);
        s->explicit_lights = desc->light_meshes != nullptr;
        if (s->explicit_lights) s->light_meshes.assign(desc->light_meshes, desc->light_meshes + desc->n_light_meshes);
        prt::setup_materials(*desc, s->mats);
        s->texs.resize(desc->n_textures);
        for (uint32_t i = 0; i < desc->n_textures; ++i) {
            const PrtTexture& t = desc->textures[i];
            DTexture& o = s->texs[i];
            o.width = t.width;
            o.height = t.height;
            o.channels = t.channels;
            o.has_data = (t.data && t.width > 0 && t.height > 0 && t.channels > 0) ? 1 : 0;
            o.offset = s->texels_lin.size();
            if (o.has_data) {
                // ImageTexture::GetPixel: colorScale * byte, SRGBToLinear for >= 3 channels, grey replicated otherwise
                double lut[256];
                for (int b = 0; b < 256; ++b) {
                    const double c = (1.0 / 255.0) * b;
                    lut[b] = (c <= 0.04045) ? c * (1. / 12.92) : std::pow((c + 0.055) * (1. / 1.055), 2.4);
                }
                const size_t npx = (size_t)t.width * t.height;
                s->texels_lin.reserve(s->texels_lin.size() + npx * 3);
                for (size_t i = 0; i < npx; ++i) {
                    const uint8_t* px = t.data + i * t.channels;
                    if (t.channels >= 3) {
                        s->texels_lin.push_back(lut[px[0]]);
                        s->texels_lin.push_back(lut[px[1]]);
                        s->texels_lin.push_back(lut[px[2]]);
                    } else {
                        const double g = (1.0 / 255.0) * px[0];
                        s->texels_lin.insert(s->texels_lin.end(), {g, g, g});
                    }
                }
            }
        }
        prt::build_light_tree(*desc, s->tris, s->mats, s->lights);
        s->device_bvh = (desc->flags & PRT_SCENE_DEVICE_BVH) && s->tris.size() >= 2;
        if (!s->device_bvh) {
            std::string err;
            const auto t0 = std::chrono::steady_clock::now();
            if (!prt::build_bvh(s->tris, s->bvh, &err)) {
                delete s;
                return fail(PRT_E_LIMIT, "prt_scene_create: " + err);
            }
            s->bvh_info.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    } catch (const std::bad_alloc&) {
        delete s;
        return fail(PRT_E_OOM, "prt_scene_create: out of host memory");
    } catch (const std::exception& e) {
        delete s;
        return fail(PRT_E_INVALID, std::string("prt_scene_create: ") + e.what());
    }
    s->last.bvh_nodes = s->bvh.nodes.size();
    s->last.bvh_depth = s->bvh.depth;
    s->bvh_info.n_nodes = s->bvh.nodes.size();
    s->bvh_info.depth = s->bvh.depth;
    *out = s;
    return PRT_OK;
}

void prt_scene_destroy(PrtScene* s) {
    if (!s) return;
    s->release();
    delete s;
}

int prt_scene_bvh_info(const PrtScene* s, PrtBvhInfo* out) {
    if (!s || !out) return fail(PRT_E_INVALID, "prt_scene_bvh_info: null argument");
    *out = s->bvh_info;
    out->node_bytes = (uint32_t)sizeof(DNode);
    out->width = PRT_BVH_WIDTH;
    out->tri_bytes = (uint32_t)sizeof(DTri);
    out->tri_stride = s->device >= 0 ? s->d.tri_stride : 0u;
    out->texture_bytes = s->device >= 0 ? (uint64_t)s->n_texel_reals * sizeof(double) : 0u;
    out->texture_footprint_bytes = s->device >= 0 ? (uint64_t)s->tex_footprint_bytes : 0u;
    out->texture_layouts = s->device >= 0 ? s->tex_layouts : 0u;
    const bool up = s->device >= 0;
    out->render_blocks_per_cu = up ? (uint32_t)s->blocks_per_cu[0] : 0u;
    out->render_blocks_wanted = up ? (uint32_t)s->blocks_wanted : 0u;
    out->lds_materials = up ? (uint32_t)s->mat_lds : 0u;
    out->lds_light_nodes = up ? (uint32_t)s->light_lds : 0u;
    out->lds_light_tris = up ? (uint32_t)s->ltri_lds : 0u;
    out->stack_need = up ? (uint32_t)s->stack_need : 0u;
    out->reserved_ = 0;
    return PRT_OK;
}

int prt_scene_light_count(const PrtScene* s, uint64_t* n) {
    if (!s || !n) return fail(PRT_E_INVALID, "prt_scene_light_count: null argument");
    *n = s->lights.tris.size();
    return PRT_OK;
}

int prt_scene_light_order(const PrtScene* s, int32_t* prims, uint64_t cap) {
    if (!s || (!prims && cap)) return fail(PRT_E_INVALID, "prt_scene_light_order: null argument");
    if (cap < s->lights.tris.size()) return fail(PRT_E_INVALID, "prt_scene_light_order: buffer too small");
    for (size_t i = 0; i < s->lights.tris.size(); ++i) prims[i] = s->lights.tris[i].prim;
    return PRT_OK;
}

static int upload_impl(PrtScene* s, int device);

// How much of the material table, the light triangles and the light tree a render kernel stages in LDS, given the
// bytes a block may spend on them (record sizes differ between the fp64 and the fp32 kernels).
static void size_tables(const PrtScene* s, int budget, size_t mat_bytes, size_t ltri_bytes, size_t lnode_bytes, int* mat, int* ltri, int* light) {
    *mat = *ltri = *light = 0;
    const bool off = dev_env("PRT_TUNE_NO_LDS") && std::atoi(dev_env("PRT_TUNE_NO_LDS"));
    if (off || s->mats.empty() || s->mats.size() * mat_bytes > 8192 || (int)(s->mats.size() * mat_bytes) > budget) return;
    *mat = (int)s->mats.size();
    budget -= *mat * (int)mat_bytes;
    if (!s->lights.tris.empty() && s->lights.tris.size() <= 32 && (int)(s->lights.tris.size() * ltri_bytes) <= budget) {
        *ltri = (int)s->lights.tris.size();
        budget -= *ltri * (int)ltri_bytes;
    }
    *light = (int)std::min<size_t>(s->lights.nodes.size(), (size_t)(std::max(budget, 0) / (int)lnode_bytes));
    if (const char* e = dev_env("PRT_TUNE_LIGHT_LDS")) *light = std::min(*light, std::max(0, std::atoi(e)));
    if (const char* e = dev_env("PRT_TUNE_LTRI_LDS")) if (!std::atoi(e)) *ltri = 0;
}

// Either the whole scene is resident afterwards, or nothing is: a failure anywhere (allocation, copy, device build)
// releases what was uploaded so far and leaves the scene in the not-uploaded state (device = -1), so that no later
// call can launch a kernel on a half-filled DScene.
int prt_scene_upload(PrtScene* s, int device) {
    if (!s) return fail(PRT_E_INVALID, "prt_scene_upload: null scene");
    const int rc = upload_impl(s, device);
    if (rc != PRT_OK) {
        const std::string keep = g_err;
        s->release();
        g_err = keep;
    }
    return rc;
}

static int upload_impl(PrtScene* s, int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(PRT_E_NO_DEVICE, "prt_scene_upload: no HIP device");
    if (device < 0 || device >= ndev) return fail(PRT_E_INVALID, "prt_scene_upload: device index out of range");
    s->release();
    PRT_HIP(hipSetDevice(device));
    s->device = device;
    s->n_uploads = 0;
    s->fail_upload_at = dev_env("PRT_TEST_FAIL_UPLOAD") ? std::atoi(dev_env("PRT_TEST_FAIL_UPLOAD")) : -1;
    hipDeviceProp_t prop;
    PRT_HIP(hipGetDeviceProperties(&prop, device));
    s->n_cu = prop.multiProcessorCount;

    // triangles in BVH leaf order
    const size_t n = s->tris.size();
    std::vector<DTri> dt(n);
    std::vector<DTriShade> ds(n);
    for (size_t i = 0; i < n; ++i) {
        const prt::HostTri& T = s->tris[s->device_bvh ? i : s->bvh.order[i]]; // device build: permuted on the GPU below
        DTri& a = dt[i];
        std::memcpy(a.n, T.normal, 24);
        a.D = T.D;
        a.A[0] = T.e1[1] * T.w[2] - T.w[1] * T.e1[2]; // e1 x w
        a.A[1] = T.e1[2] * T.w[0] - T.w[2] * T.e1[0];
        a.A[2] = T.e1[0] * T.w[1] - T.w[0] * T.e1[1];
        a.B[0] = T.w[1] * T.e0[2] - T.e0[1] * T.w[2]; // w x e0
        a.B[1] = T.w[2] * T.e0[0] - T.e0[2] * T.w[0];
        a.B[2] = T.w[0] * T.e0[1] - T.e0[0] * T.w[1];
        a.a0 = T.v[0][0] * a.A[0] + T.v[0][1] * a.A[1] + T.v[0][2] * a.A[2];
        a.b0 = T.v[0][0] * a.B[0] + T.v[0][1] * a.B[1] + T.v[0][2] * a.B[2];
        DTriShade& b = ds[i];
        std::memset(&b, 0, sizeof(b));
        std::memcpy(b.tangent, T.tangent, 24);
        std::memcpy(b.uv0, T.uv[0], 16);
        std::memcpy(b.uv1, T.uv[1], 16);
        std::memcpy(b.uv2, T.uv[2], 16);
        b.material = T.material;
        b.prim = T.prim;
    }
    DScene& d = s->d;
    std::memset(&d, 0, sizeof(d));
    int rc;
    // packed records for scenes the caches hold, one record per 128-byte line for scenes that stream from HBM
    uint32_t stride = (uint64_t)n * sizeof(DTri) > PRT_TRI_PADDED_ABOVE ? 128u : (uint32_t)sizeof(DTri);
    if (const char* e = dev_env("PRT_TUNE_TRI_STRIDE")) stride = std::atoi(e) == 128 ? 128u : (uint32_t)sizeof(DTri);
    if (sizeof(DTri) > 96) stride = (uint32_t)sizeof(DTri);
    d.tri_stride = stride;
    auto up_tris = [&](const DTri** out) -> int { // host records (packed) -> device records `stride` bytes apart
        if (stride == sizeof(DTri)) return s->up(dt, out);
        void* p = nullptr;
        PRT_HIP(hipMalloc(&p, std::max<size_t>(n * (size_t)stride, 256)));
        s->allocs.push_back(p);
        if (n) {
            // the padded array is laid out on the host and goes up in one copy (a 2-D copy of millions of 96-byte rows
            // from pageable memory is served row by row)
            std::vector<char> padded;
            try {
                padded.assign(n * (size_t)stride, 0);
            } catch (const std::bad_alloc&) {
                return fail(PRT_E_OOM, "prt_scene_upload: out of host memory for the padded triangle records");
            }
            for (size_t i = 0; i < n; ++i) std::memcpy(padded.data() + i * (size_t)stride, &dt[i], sizeof(DTri));
            PRT_HIP(hipMemcpy(p, padded.data(), padded.size(), hipMemcpyHostToDevice));
        }
        *out = static_cast<const DTri*>(p);
        return PRT_OK;
    };
    if (s->device_bvh) {
        std::vector<prt::PrimBox> pb;
        float box_origin[3];
        prt::prim_boxes(s->tris, pb, box_origin);
        prt::DeviceBVH db;
        std::string err;
        if (!prt::build_bvh_device(pb.data(), box_origin, n, db, &err)) return fail(PRT_E_HIP, "prt_scene_upload: " + err);
        s->allocs.push_back(db.d_nodes);
        d.nodes = db.d_nodes;
        if (dev_env("PRT_VALIDATE_BVH")) { // tests: check the device-built tree on the host before any ray visits it
            std::vector<DNode> hn(db.n_nodes);
            PRT_HIP(hipMemcpy(hn.data(), db.d_nodes, (size_t)db.n_nodes * sizeof(DNode), hipMemcpyDeviceToHost));
            if (!prt::validate_nodes(hn.data(), hn.size(), n, &err)) {
                (void)hipFree(db.d_order);
                return fail(PRT_E_LIMIT, "prt_scene_upload: device-built BVH is malformed: " + err);
            }
        }
        // records go up in description order and are permuted into BVH leaf order in HBM
        const DTri* t_in = nullptr;
        const DTriShade* s_in = nullptr;
        void *t_out = nullptr, *s_out = nullptr;
        const size_t mark = s->allocs.size();
        rc = s->up(dt, &t_in);
        if (!rc) rc = s->up(ds, &s_in);
        hipError_t e = hipSuccess;
        if (!rc) {
            e = hipMalloc(&t_out, n * (size_t)stride);
            if (e == hipSuccess) { s->allocs.push_back(t_out); e = hipMalloc(&s_out, n * sizeof(DTriShade)); }
            if (e == hipSuccess) {
                s->allocs.push_back(s_out);
                prt::launch_gather_tris(t_in, s_in, db.d_order, (uint32_t)n, t_out, stride,
                                        static_cast<DTriShade*>(s_out), nullptr);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipDeviceSynchronize();
            }
        }
        (void)hipFree(db.d_order);
        // drop the two staging copies (they sit at allocs[mark], allocs[mark+1] when their upload succeeded)
        for (const void* p : {static_cast<const void*>(t_in), static_cast<const void*>(s_in)})
            if (p) {
                (void)hipFree(const_cast<void*>(p));
                s->allocs.erase(std::find(s->allocs.begin() + mark, s->allocs.end(), const_cast<void*>(p)));
            }
        if (rc) return rc;
        if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? PRT_E_OOM : PRT_E_HIP, std::string("prt_scene_upload: ") + hipGetErrorString(e));
        d.tris = static_cast<const DTri*>(t_out);
        d.shade = static_cast<const DTriShade*>(s_out);
        s->bvh.depth = db.depth;
        s->bvh.coord_scale = db.coord_scale;
        for (int a = 0; a < 3; ++a) {
            s->bvh.grid_origin[a] = db.grid_origin[a];
            s->bvh.grid_step[a] = db.grid_step[a];
        }
        s->last.bvh_nodes = s->bvh_info.n_nodes = db.n_nodes;
        s->last.bvh_depth = s->bvh_info.depth = db.depth;
        s->bvh_info.built_on_device = 1;
        s->bvh_info.build_ms = db.ms_total;
        s->bvh_info.sort_ms = db.ms_sort;
        s->bvh_info.tree_ms = db.ms_tree;
        s->bvh_info.split_ms = db.ms_split;
    } else {
        if ((rc = s->up(s->bvh.nodes, &d.nodes))) return rc;
        if ((rc = up_tris(&d.tris))) return rc;
        if ((rc = s->up(ds, &d.shade))) return rc;
    }
    if ((rc = s->up(s->mats, &d.materials))) return rc;
    {
        // Device layout of the texels, per SCENE (DScene::tex_compact): BILINEAR FOOTPRINTS — per texel cell the four taps
        // Value() blends, 16 reals = one 128-byte line per lookup (bathroom2 -1.8 %) at 5.3x the bytes — while all the scene's
        // footprints fit PRT_TEX_FOOTPRINT_BUDGET (256 MiB of fp64 records = the size of the Infinity Cache; the fp32 fast mode
        // adds half as much again when it is used); else the plain row-major texel arrays (3 reals per texel, a lookup touches
        // two to four lines): a 4096^2 texture is 403 MB instead of 2.1 GB.  Same doubles, same blend, either way.
        std::vector<DTexture> qt(s->texs);
        std::vector<double> quads;
        size_t budget = PRT_TEX_FOOTPRINT_BUDGET;
        if (const char* e = dev_env("PRT_TUNE_TEX_BUDGET")) budget = (size_t)std::strtoull(e, nullptr, 10);
        size_t cells_all = 0;
        for (const DTexture& t : s->texs) cells_all += t.has_data ? (size_t)t.width * t.height : 0;
        const bool compact = cells_all * 16 * sizeof(double) > budget;
        const size_t per_cell = compact ? 3 : 16;
        try {
            quads.assign(cells_all * per_cell, 0.0);
        } catch (const std::bad_alloc&) {
            return fail(PRT_E_OOM, "prt_scene_upload: out of host memory for the texel arrays");
        }
        size_t at = 0;
        for (size_t i = 0; i < s->texs.size(); ++i) {
            const DTexture& t = s->texs[i];
            qt[i].offset = at;
            if (!t.has_data) continue;
            const double* px = s->texels_lin.data() + t.offset;
            if (compact) {
                const size_t n3 = (size_t)t.width * t.height * 3;
                std::memcpy(quads.data() + at, px, n3 * sizeof(double));
                at += n3;
                continue;
            }
            for (int y0 = 0; y0 < t.height; ++y0)
                for (int x0 = 0; x0 < t.width; ++x0) {
                    const int x1 = std::min(x0 + 1, t.width - 1), y1 = std::min(y0 + 1, t.height - 1); // Texture.cpp:35-36
                    double* o = quads.data() + at;
                    const int tap[4][2] = {{x0, y0}, {x1, y0}, {x0, y1}, {x1, y1}};
                    for (int k = 0; k < 4; ++k) std::memcpy(o + 3 * k, px + ((size_t)tap[k][1] * t.width + tap[k][0]) * 3, 24);
                    at += 16;
                }
        }
        d.tex_compact = compact ? 1u : 0u;
        s->tex_footprint_bytes = compact ? 0 : cells_all * 16 * sizeof(double);
        s->tex_layouts = cells_all == 0 ? 0u : (compact ? 2u : 1u);
        s->n_texel_reals = quads.size();
        if ((rc = s->up(qt, &d.textures))) return rc;
        if ((rc = s->up(quads, &d.texels_lin))) return rc;
    }
    if ((rc = s->up(s->lights.nodes, &d.light_nodes))) return rc;
    d.light_tab = nullptr;
    if (!s->lights.tab.empty() && (rc = s->up(s->lights.tab, &d.light_tab))) return rc;
    if ((rc = s->up(s->lights.tris, &d.light_tris))) return rc;
    d.light_root = s->lights.root;
    d.n_lights = (int32_t)s->lights.tris.size();
    d.light_area = s->lights.area;
    d.n_nodes = s->device_bvh ? (uint32_t)s->bvh_info.n_nodes : (uint32_t)s->bvh.nodes.size();
    d.n_tris = (uint32_t)n;
    d.slab_scale = s->bvh.coord_scale;
    for (int a = 0; a < 3; ++a) {
        d.grid_origin[a] = s->bvh.grid_origin[a];
        d.grid_step[a] = s->bvh.grid_step[a];
    }
    {
        // box coordinates reach the slab test relative to the grid origin: what bounds its rounding is the grid's extent
        double e = 0.0;
        for (int a = 0; a < 3; ++a) e = std::max(e, 65535.0 * (double)s->bvh.grid_step[a]);
        d.slab_scale = std::nextafter((float)e, std::numeric_limits<float>::infinity());
    }
    for (PrtScene::CallSlot& q : s->slots) {
        PRT_HIP(hipMalloc(reinterpret_cast<void**>(&q.d_ctr), sizeof(DCounters)));
        PRT_HIP(hipMemset(q.d_ctr, 0, sizeof(DCounters)));
        PRT_HIP(hipEventCreate(&q.ev0));
        PRT_HIP(hipEventCreate(&q.ev1));
        PRT_HIP(hipEventCreateWithFlags(&q.done, hipEventDisableTiming));
    }
    s->feat = 0;
    for (const DMaterial& m : s->mats) {
        if (m.texture >= 0) s->feat |= 1;
        if (m.type == PRT_MAT_PHONG) s->feat |= 2;
        if (m.type == PRT_MAT_COOKTORRANCE) s->feat |= 4;
    }
    if (const char* e = dev_env("PRT_TUNE_FEAT")) s->feat |= std::atoi(e); // developer: force a larger permutation
    s->feat = prt::render_permutation(s->feat);
    // Small read-only tables of the shading code live in LDS (LLDS kernels): every read of them is otherwise a
    // texture-addresser instruction, and the light tree's descent is a chain of dependent reads.  In order of
    // value per byte: the material table (must fit, <= 8 KB), all light triangles if there are at most 32, then
    // as many top levels of the light tree (breadth-first numbering) as the remaining budget holds.
    // Measured: materials cornell +2.5 %, bathroom2 +3 %, veach-mis +2 %; light tree veach-mis +5 %.
    // LDS traversal stacks of K3 are sized from what THIS tree can need (tree_stack_need), not from the builders' bound
    {
        int need = PRT_STACK_DEPTH;
        if (!s->bvh_info.built_on_device) need = s->bvh.stack_need;
        else if (d.n_nodes <= (1u << 21)) { // device-built: the nodes come back once for the count (at most 128 MB; larger trees keep the bound)
            std::vector<DNode> hn(d.n_nodes);
            PRT_HIP(hipMemcpy(hn.data(), d.nodes, (size_t)d.n_nodes * sizeof(DNode), hipMemcpyDeviceToHost));
            need = prt::tree_stack_need(hn.data(), hn.size());
        }
        s->stack_need = std::min(std::max(need, 1), PRT_STACK_DEPTH);
        s->d_k3 = d;
        s->d_nodes_shallow = nullptr;
        s->n_nodes_shallow = 0;
        // The fp64 render kernels keep STATIC stacks of PRT_STACK_DEPTH entries per lane (they are register-limited to three
        // blocks per CU; a run-time depth cost them 1.2 %), so a shallower collapse of the tree frees them no LDS: they
        // always traverse the full tree, and their table budget is what those static stacks leave.  (The fp32 kernels size
        // their stacks per launch and do take the 32-entry collapse of a deep host-built tree: upload_f32_tables.)
        s->stack_depth = PRT_STACK_DEPTH;
        if (dev_env("PRT_TUNE_VERBOSE")) std::fprintf(stderr, "[prt] tree needs %d stack entries\n", s->stack_need);
    }
    size_tables(s, prt::render_lds_budget(s->feat, s->stack_depth), sizeof(DMaterial), sizeof(DLightTri), sizeof(DLightNode),
                &s->mat_lds, &s->ltri_lds, &s->light_lds);
    static_assert(sizeof(DLightNode) == 16 && sizeof(DLightTri) % 16 == 0, "LDS staging copies 16-byte pieces");
    const size_t tables = prt::render_table_bytes(s->light_lds, s->mat_lds, s->ltri_lds);
    const bool pad = d.tri_stride == PRT_TRI_PAD_STRIDE(double) && sizeof(DTri) != PRT_TRI_PAD_STRIDE(double);
    s->blocks_per_cu[0] = prt::render_blocks_per_cu(false, s->feat, tables, s->stack_depth, pad, !s->lights.tab.empty() || s->d.tex_compact != 0);
    s->blocks_wanted = tables != 0 ? prt::render_blocks_per_cu(false, s->feat, 0, s->stack_depth, pad, !s->lights.tab.empty() || s->d.tex_compact != 0) : s->blocks_per_cu[0];
    if (s->blocks_per_cu[0] < s->blocks_wanted) {
        // the tables would cost the production kernel a resident block (the budget is an estimate; the occupancy query is
        // the truth): a block per CU is worth far more than the tables — render without them
        s->mat_lds = s->ltri_lds = s->light_lds = 0;
        s->blocks_per_cu[0] = s->blocks_wanted;
    }
    const size_t tables_used = prt::render_table_bytes(s->light_lds, s->mat_lds, s->ltri_lds);
    s->blocks_per_cu[1] = prt::render_blocks_per_cu(true, s->feat, tables_used, s->stack_depth, pad, !s->lights.tab.empty() || s->d.tex_compact != 0);
    if (dev_env("PRT_TUNE_VERBOSE"))
        std::fprintf(stderr, "[prt] fp64 render kernels: %d blocks per CU, LDS tables: %d materials, %d light triangles, %d light nodes\n",
                     s->blocks_per_cu[0], s->mat_lds, s->ltri_lds, s->light_lds);
    return PRT_OK;
}

int prt_scene_update_vertices(PrtScene* s, const double* vertices, const double* normals) {
    if (!s || (!vertices && !s->tris.empty())) return fail(PRT_E_INVALID, "prt_scene_update_vertices: null argument");
    for (size_t i = 0; i < s->tris.size() * 9; ++i)
        if (!(std::fabs(vertices[i]) <= 1e18)) return fail(PRT_E_INVALID, "prt_scene_update_vertices: vertex coordinate is not finite (or beyond 1e18)");
    try {
        prt::update_triangles(vertices, normals, s->tris);
        PrtSceneDesc d;
        std::memset(&d, 0, sizeof(d));
        d.n_tris = s->tris.size();
        d.n_meshes = (uint32_t)s->mesh_mat.size();
        d.mesh_first_tri = s->mesh_first.data();
        d.mesh_material = s->mesh_mat.data();
        static const int32_t none = 0;
        d.light_meshes = s->explicit_lights ? (s->light_meshes.empty() ? &none : s->light_meshes.data()) : nullptr;
        d.n_light_meshes = (uint32_t)s->light_meshes.size();
        s->lights = prt::LightTree();
        prt::build_light_tree(d, s->tris, s->mats, s->lights); // light areas and the CDF order follow the geometry
        if (s->device >= 0 && s->tris.size() >= 2) {
            // moving geometry: the tree is rebuilt on the GPU (milliseconds; no refit needed, no quality decay)
            s->device_bvh = true;
        } else if (!s->device_bvh) {
            std::string err;
            const auto t0 = std::chrono::steady_clock::now();
            if (!prt::build_bvh(s->tris, s->bvh, &err)) return fail(PRT_E_LIMIT, "prt_scene_update_vertices: " + err);
            s->bvh_info.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            s->last.bvh_nodes = s->bvh_info.n_nodes = s->bvh.nodes.size();
            s->last.bvh_depth = s->bvh_info.depth = s->bvh.depth;
        }
    } catch (const std::bad_alloc&) {
        return fail(PRT_E_OOM, "prt_scene_update_vertices: out of host memory");
    }
    if (s->device >= 0) return prt_scene_upload(s, s->device);
    return PRT_OK;
}

// fp32 fast mode: the float tables are derived from the resident fp64 ones the first time they are asked for
// (synchronous; the scene then holds both).  Triangle and shading records and texels are converted on the device —
// they already are in BVH leaf order there, whichever builder made the tree — the small tables on the host.
static int ensure_f32_impl(PrtScene* s);
// All or nothing, like the upload: a failure frees what this call allocated and leaves the scene without fp32 tables.
static int ensure_f32(PrtScene* s) {
    if (s->f32_ready) return PRT_OK;
    const size_t mark = s->allocs.size();
    const int rc = ensure_f32_impl(s);
    if (rc != PRT_OK) {
        const std::string keep = g_err;
        for (size_t i = mark; i < s->allocs.size(); ++i) (void)hipFree(s->allocs[i]);
        s->allocs.resize(mark);
        std::memset(&s->d32, 0, sizeof(s->d32));
        g_err = keep;
    }
    return rc;
}
static int ensure_f32_impl(PrtScene* s) {
    const DScene& d = s->d;
    DSceneT<float>& f = s->d32;
    std::memset(&f, 0, sizeof(f));
    const size_t n = d.n_tris;
    const uint32_t stride = d.tri_stride == sizeof(DTriT<double>) ? (uint32_t)sizeof(DTriT<float>) : PRT_TRI_PAD_STRIDE(float);
    void *t = nullptr, *sh = nullptr, *tx = nullptr;
    PRT_HIP(hipMalloc(&t, std::max<size_t>(n * (size_t)stride, 256)));
    s->allocs.push_back(t);
    PRT_HIP(hipMalloc(&sh, std::max<size_t>(n * sizeof(DTriShadeT<float>), 256)));
    s->allocs.push_back(sh);
    PRT_HIP(hipMalloc(&tx, std::max<size_t>(s->n_texel_reals * sizeof(float), 256)));
    s->allocs.push_back(tx);
    prt32::launch_convert_tris(d.tris, d.tri_stride, (uint32_t)n, t, stride, nullptr);
    prt32::launch_convert_shade(d.shade, (uint32_t)n, static_cast<DTriShadeT<float>*>(sh), nullptr);
    prt32::launch_convert_reals(d.texels_lin, s->n_texel_reals, static_cast<float*>(tx), nullptr);
    PRT_HIP(hipGetLastError());
    std::vector<DMaterialT<float>> mats(s->mats.size());
    for (size_t i = 0; i < mats.size(); ++i) {
        const DMaterial& a = s->mats[i];
        DMaterialT<float>& o = mats[i];
        std::memset(&o, 0, sizeof(o));
        o.type = a.type; o.texture = a.texture;
        conv_arr(o.kd, a.kd, 3); conv_arr(o.ks, a.ks, 3); conv_arr(o.emission, a.emission, 3);
        conv_arr(o.eta, a.eta, 3); conv_arr(o.k, a.k, 3);
        o.ns = (float)a.ns; o.pkd = (float)a.pkd; o.pks = (float)a.pks;
        o.alpha_x = (float)a.alpha_x; o.alpha_y = (float)a.alpha_y;
        o.has_emission = a.has_emission; o.skip_light_sampling = a.skip_light_sampling;
        o.inv_ns1 = (float)a.inv_ns1; o.spec_scale = (float)a.spec_scale;
    }
    std::vector<DLightNodeT<float>> ln(s->lights.nodes.size());
    for (size_t i = 0; i < ln.size(); ++i) {
        std::memset(&ln[i], 0, sizeof(ln[i]));
        ln[i].left_area = (float)s->lights.nodes[i].left_area;
        ln[i].left = s->lights.nodes[i].left;
        ln[i].right = s->lights.nodes[i].right;
    }
    std::vector<DLightTriT<float>> lt(s->lights.tris.size());
    for (size_t i = 0; i < lt.size(); ++i) {
        const DLightTri& a = s->lights.tris[i];
        DLightTriT<float>& o = lt[i];
        std::memset(&o, 0, sizeof(o));
        conv_arr(o.v0, a.v0, 3); conv_arr(o.v1, a.v1, 3); conv_arr(o.v2, a.v2, 3); conv_arr(o.n, a.n, 3);
        o.area = (float)a.area; o.material = a.material; o.prim = a.prim; o.pdf = (float)a.pdf;
    }
    int rc;
    const int keep_fail = s->fail_upload_at;
    s->fail_upload_at = -1;
    if ((rc = s->up(mats, &f.materials)) || (rc = s->up(ln, &f.light_nodes)) || (rc = s->up(lt, &f.light_tris))) {
        s->fail_upload_at = keep_fail;
        return rc;
    }
    s->fail_upload_at = keep_fail;
    PRT_HIP(hipDeviceSynchronize());
    f.nodes = d.nodes;
    f.tris = static_cast<const DTriT<float>*>(t);
    f.shade = static_cast<const DTriShadeT<float>*>(sh);
    f.textures = d.textures;
    f.texels_lin = static_cast<const float*>(tx);
    f.light_root = d.light_root;
    f.light_tab = d.light_tab; // thresholds are floats in either mode
    f.n_lights = d.n_lights;
    f.light_area = (float)d.light_area;
    f.n_nodes = d.n_nodes;
    f.n_tris = d.n_tris;
    f.slab_scale = d.slab_scale;
    for (int a = 0; a < 3; ++a) {
        f.grid_origin[a] = d.grid_origin[a];
        f.grid_step[a] = d.grid_step[a];
    }
    f.tri_stride = stride;
    f.tex_compact = d.tex_compact;
    // The fp32 kernels have the registers for a fourth wave per SIMD; whether the LDS has room for a fourth block per CU
    // is decided by the traversal stacks: 32 entries per lane (32 KB per block) leave it, the builders' bound of
    // PRT_STACK_DEPTH does not.  Most trees need far fewer entries than that bound (tree_stack_need).
    {
        int need = s->stack_need; // of the tree the fp64 K3 traverses
        if (need > PRT_STACK_SHALLOW && !s->d_nodes_shallow && !s->bvh_info.built_on_device && !s->bvh.nodes_shallow.empty()) {
            // the same binary tree collapsed for 32 entries (same leaf order: the records above fit both): the fp32 kernels
            // have the registers for a fourth block per CU, which 32-entry stacks leave the LDS for
            const int keep = s->fail_upload_at;
            s->fail_upload_at = -1;
            rc = s->up(s->bvh.nodes_shallow, &s->d_nodes_shallow);
            s->fail_upload_at = keep;
            if (rc) return rc;
            s->n_nodes_shallow = (uint32_t)s->bvh.nodes_shallow.size();
        }
        if (s->d_nodes_shallow) {
            f.nodes = s->d_nodes_shallow;
            f.n_nodes = s->n_nodes_shallow;
            need = std::min(need, PRT_STACK_SHALLOW);
        }
        s->stack_depth32 = need <= PRT_STACK_SHALLOW ? PRT_STACK_SHALLOW : PRT_STACK_DEPTH;
        if (const char* e = dev_env("PRT_TUNE_STACK32")) s->stack_depth32 = std::max(need, std::min(PRT_STACK_DEPTH, std::atoi(e))); // developer: smaller stacks when the tree allows
        if (dev_env("PRT_TUNE_VERBOSE")) std::fprintf(stderr, "[prt] fp32 tables: tree needs %d stack entries, using %d\n", need, s->stack_depth32);
    }
    size_tables(s, prt32::render_lds_budget(s->feat, s->stack_depth32), sizeof(DMaterialT<float>), sizeof(DLightTriT<float>),
                sizeof(DLightNodeT<float>), &s->mat_lds32, &s->ltri_lds32, &s->light_lds32);
    const size_t tables = prt32::render_table_bytes(s->light_lds32, s->mat_lds32, s->ltri_lds32);
    const bool pad = stride == PRT_TRI_PAD_STRIDE(float) && sizeof(DTriT<float>) != PRT_TRI_PAD_STRIDE(float);
    s->blocks_per_cu32[0] = prt32::render_blocks_per_cu(false, s->feat, tables, s->stack_depth32, pad, !s->lights.tab.empty() || s->d.tex_compact != 0);
    s->blocks_per_cu32[1] = prt32::render_blocks_per_cu(true, s->feat, tables, s->stack_depth32, pad, !s->lights.tab.empty() || s->d.tex_compact != 0);
    if (dev_env("PRT_TUNE_VERBOSE"))
        std::fprintf(stderr, "[prt] fp32 render kernels: %d blocks per CU, stacks %d, LDS tables: %d materials, %d light triangles, %d light nodes\n",
                     s->blocks_per_cu32[0], s->stack_depth32, s->mat_lds32, s->ltri_lds32, s->light_lds32);
    s->f32_ready = true;
    return PRT_OK;
}

// After the counters of a call slot have been zeroed: the pointers K3 reads from them (DCounters::pixel_list / trace).
static int set_slot_pointers(PrtScene*, PrtScene::CallSlot& q, hipStream_t st, unsigned, const void* pixel_list, void* trace) {
    if (!pixel_list && !trace) return PRT_OK; // zeroed already
    const void* ptrs[2] = {pixel_list, trace};
    PRT_HIP(hipMemcpyAsync(reinterpret_cast<char*>(q.d_ctr) + offsetof(DCounters, pixel_list), ptrs, sizeof(ptrs), hipMemcpyHostToDevice, st));
    return PRT_OK;
}

static int require_uploaded(PrtScene* s, const char* who) {
    if (!s) return fail(PRT_E_INVALID, std::string(who) + ": null scene");
    if (s->device < 0) return fail(PRT_E_NO_DEVICE, std::string(who) + ": scene is not uploaded to a HIP device (no CPU path exists)");
    PRT_HIP(hipSetDevice(s->device));
    return PRT_OK;
}

int prt_trace_closest_device(PrtScene* s, const void* d_rays, size_t n, void* d_hits, int count_work, void* stream) {
    return prt_trace_closest_device_prec(s, d_rays, n, d_hits, count_work, PRT_PRECISION_F64, stream);
}

static int trace_closest_device(PrtScene* s, const void* d_rays, size_t n, void* d_hits, int count_work, int precision, void* stream, bool sorted);
int prt_trace_closest_device_prec(PrtScene* s, const void* d_rays, size_t n, void* d_hits, int count_work, int precision,
                                  void* stream) {
    return trace_closest_device(s, d_rays, n, d_hits, count_work, precision, stream, false);
}
int prt_trace_closest_sorted_device(PrtScene* s, const void* d_rays, size_t n, void* d_hits, int count_work, int precision,
                                    void* stream) {
    return trace_closest_device(s, d_rays, n, d_hits, count_work, precision, stream, true);
}
static int trace_closest_device(PrtScene* s, const void* d_rays, size_t n, void* d_hits, int count_work, int precision, void* stream, bool sorted) {
    int rc = require_uploaded(s, "prt_trace_closest_device");
    if (rc) return rc;
    if (sorted && n > 0xffffffffull) return fail(PRT_E_INVALID, "prt_trace_closest_sorted_device: more than 2^32 - 1 rays in one batch");
    if (n && (!d_rays || !d_hits)) return fail(PRT_E_INVALID, "prt_trace_closest_device: null buffer");
    if (precision != PRT_PRECISION_F64 && precision != PRT_PRECISION_F32) return fail(PRT_E_INVALID, "prt_trace_closest_device: unsupported precision");
    if (precision == PRT_PRECISION_F32 && (rc = ensure_f32(s))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t we;
    PrtScene::CallSlot& q = *s->next_slot(st, &we);
    PRT_HIP(we);
    PRT_HIP(hipMemsetAsync(q.d_ctr, 0, sizeof(DCounters), st));
    const uint32_t* d_perm = nullptr;
    if (sorted && n > 1) {
        std::string err;
        const size_t need = prt::ray_sort_scratch_bytes(n, &err);
        if (!need) return fail(PRT_E_HIP, err);
        if (!s->sort_done) PRT_HIP(hipEventCreateWithFlags(&s->sort_done, hipEventDisableTiming));
        else PRT_HIP(hipStreamWaitEvent(st, s->sort_done, 0)); // the scratch is the previous sorted call's until that call has ended
        if (s->sort_cap < need) {
            PRT_HIP(hipEventSynchronize(s->sort_done)); // (never recorded: returns at once)
            if (s->d_sort) (void)hipFree(s->d_sort);    // hipFree waits for the device
            s->d_sort = nullptr;
            s->sort_cap = 0;
            if (hipMalloc(&s->d_sort, need) != hipSuccess) return fail(PRT_E_OOM, "prt_trace_closest_sorted_device: hipMalloc of the sort scratch failed");
            s->sort_cap = need;
        }
    }
    PRT_HIP(hipEventRecord(q.ev0, st)); // (the sort is inside the timed region: kernel_ms is keys + sort + trace)
    if (sorted && n > 1) {
        std::string err;
        d_perm = prt::ray_sort(static_cast<const PrtRay*>(d_rays), n, s->d.grid_origin, s->d.grid_step, s->d_sort, s->sort_cap, st, &err);
        if (!d_perm) return fail(PRT_E_HIP, err);
    }
    if (precision == PRT_PRECISION_F32)
        prt32::launch_trace(s->d32, static_cast<const PrtRay*>(d_rays), n, static_cast<PrtHit*>(d_hits), q.d_ctr, count_work != 0,
                            s->n_cu, st, d_perm);
    else
        prt::launch_trace(s->d, static_cast<const PrtRay*>(d_rays), n, static_cast<PrtHit*>(d_hits), q.d_ctr, count_work != 0,
                          s->n_cu, st, d_perm);
    PRT_HIP(hipGetLastError());
    PRT_HIP(hipEventRecord(q.ev1, st));
    PRT_HIP(hipEventRecord(q.done, st));
    if (d_perm) PRT_HIP(hipEventRecord(s->sort_done, st));
    q.timed = true;
    q.counted = count_work != 0;
    q.samples = 0;
    return PRT_OK;
}

int prt_trace_closest(PrtScene* s, const PrtRay* rays, size_t n, PrtHit* hits, int count_work) {
    int rc = require_uploaded(s, "prt_trace_closest");
    if (rc) return rc;
    if (n == 0) return PRT_OK;
    if (!rays || !hits) return fail(PRT_E_INVALID, "prt_trace_closest: null buffer");
    void *dr = nullptr, *dh = nullptr;
    PRT_HIP(hipMalloc(&dr, n * sizeof(PrtRay)));
    hipError_t e = hipMalloc(&dh, n * sizeof(PrtHit));
    if (e != hipSuccess) {
        (void)hipFree(dr);
        return fail(PRT_E_OOM, "prt_trace_closest: hipMalloc failed");
    }
    rc = PRT_OK;
    do {
        if (hipMemcpy(dr, rays, n * sizeof(PrtRay), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(PRT_E_HIP, "prt_trace_closest: H2D copy failed"); break; }
        rc = prt_trace_closest_device(s, dr, n, dh, count_work, nullptr);
        if (rc) break;
        e = hipDeviceSynchronize();
        if (e != hipSuccess) { rc = fail(PRT_E_HIP, std::string("prt_trace_closest: ") + hipGetErrorString(e)); break; }
        if (hipMemcpy(hits, dh, n * sizeof(PrtHit), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(PRT_E_HIP, "prt_trace_closest: D2H copy failed"); break; }
    } while (0);
    (void)hipFree(dr);
    (void)hipFree(dh);
    return rc;
}

int prt_sample_lights(PrtScene* s, const double* origins, size_t n, uint64_t seed, PrtLightSample* out) {
    int rc = require_uploaded(s, "prt_sample_lights");
    if (rc) return rc;
    if (n == 0) return PRT_OK;
    if (!origins || !out) return fail(PRT_E_INVALID, "prt_sample_lights: null buffer");
    if (s->d.n_lights == 0) return fail(PRT_E_INVALID, "prt_sample_lights: scene has no emissive mesh");
    void *dorg = nullptr, *dout = nullptr;
    PRT_HIP(hipMalloc(&dorg, n * 3 * sizeof(double)));
    if (hipMalloc(&dout, n * sizeof(PrtLightSample)) != hipSuccess) {
        (void)hipFree(dorg);
        return fail(PRT_E_OOM, "prt_sample_lights: hipMalloc failed");
    }
    rc = PRT_OK;
    do {
        if (hipMemcpy(dorg, origins, n * 3 * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(PRT_E_HIP, "prt_sample_lights: H2D copy failed"); break; }
        prt::launch_sample_lights(s->d, static_cast<const double*>(dorg), n, seed, static_cast<PrtLightSample*>(dout), nullptr);
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { rc = fail(PRT_E_HIP, std::string("prt_sample_lights: ") + hipGetErrorString(e)); break; }
        if (hipMemcpy(out, dout, n * sizeof(PrtLightSample), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(PRT_E_HIP, "prt_sample_lights: D2H copy failed"); break; }
    } while (0);
    (void)hipFree(dorg);
    (void)hipFree(dout);
    return rc;
}

namespace {
// Device staging for the small host-buffer test hooks: copies inputs up, frees everything on destruction.
struct HookBufs {
    std::vector<void*> p;
    ~HookBufs() {
        for (void* q : p) (void)hipFree(q);
    }
    void* up(const void* src, size_t bytes) { // nullptr in -> nullptr out
        if (!src) return nullptr;
        void* d = nullptr;
        if (hipMalloc(&d, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
        p.push_back(d);
        if (hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        return d;
    }
    void* out(size_t bytes) {
        void* d = nullptr;
        if (hipMalloc(&d, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr;
        p.push_back(d);
        return d;
    }
};
} // namespace

int prt_material_eval(PrtScene* s, int32_t material, size_t n, const double* wi, const double* wo, const double* uv,
                      uint64_t seed, double* f) {
    int rc = require_uploaded(s, "prt_material_eval");
    if (rc) return rc;
    if (material < 0 || (size_t)material >= s->mats.size()) return fail(PRT_E_INVALID, "prt_material_eval: material index out of range");
    if (n == 0) return PRT_OK;
    if (!wi || !wo || !f) return fail(PRT_E_INVALID, "prt_material_eval: null buffer");
    HookBufs b;
    const double *dwi = static_cast<const double*>(b.up(wi, n * 24)), *dwo = static_cast<const double*>(b.up(wo, n * 24));
    const double* duv = static_cast<const double*>(b.up(uv, n * 16));
    double* df = static_cast<double*>(b.out(n * 24));
    if (!dwi || !dwo || !df || (uv && !duv)) return fail(PRT_E_OOM, "prt_material_eval: device staging failed");
    prt::launch_material_eval(s->d, material, dwi, dwo, duv, n, seed, df, nullptr);
    PRT_HIP(hipDeviceSynchronize());
    PRT_HIP(hipMemcpy(f, df, n * 24, hipMemcpyDeviceToHost));
    return PRT_OK;
}

int prt_material_scatter(PrtScene* s, int32_t material, size_t n, const double* rd, const double* normal, const double* tangent,
                         const double* uv, uint64_t seed, double* wi_world, double* attenuation, int32_t* ok) {
    int rc = require_uploaded(s, "prt_material_scatter");
    if (rc) return rc;
    if (material < 0 || (size_t)material >= s->mats.size()) return fail(PRT_E_INVALID, "prt_material_scatter: material index out of range");
    if (n == 0) return PRT_OK;
    if (!rd || !normal || !tangent || !wi_world || !attenuation || !ok) return fail(PRT_E_INVALID, "prt_material_scatter: null buffer");
    HookBufs b;
    const double* drd = static_cast<const double*>(b.up(rd, n * 24));
    const double* duv = static_cast<const double*>(b.up(uv, n * 16));
    double *dwi = static_cast<double*>(b.out(n * 24)), *datt = static_cast<double*>(b.out(n * 24));
    int32_t* dok = static_cast<int32_t*>(b.out(n * 4));
    if (!drd || !dwi || !datt || !dok || (uv && !duv)) return fail(PRT_E_OOM, "prt_material_scatter: device staging failed");
    prt::launch_material_scatter(s->d, material, drd, normal, tangent, duv, n, seed, dwi, datt, dok, nullptr);
    PRT_HIP(hipDeviceSynchronize());
    PRT_HIP(hipMemcpy(wi_world, dwi, n * 24, hipMemcpyDeviceToHost));
    PRT_HIP(hipMemcpy(attenuation, datt, n * 24, hipMemcpyDeviceToHost));
    PRT_HIP(hipMemcpy(ok, dok, n * 4, hipMemcpyDeviceToHost));
    return PRT_OK;
}

int prt_texture_value(PrtScene* s, int32_t texture, size_t n, const double* uv, double* rgb) {
    int rc = require_uploaded(s, "prt_texture_value");
    if (rc) return rc;
    if (texture < 0 || (size_t)texture >= s->texs.size()) return fail(PRT_E_INVALID, "prt_texture_value: texture index out of range");
    if (n == 0) return PRT_OK;
    if (!uv || !rgb) return fail(PRT_E_INVALID, "prt_texture_value: null buffer");
    HookBufs b;
    const double* duv = static_cast<const double*>(b.up(uv, n * 16));
    double* d = static_cast<double*>(b.out(n * 24));
    if (!duv || !d) return fail(PRT_E_OOM, "prt_texture_value: device staging failed");
    prt::launch_texture_value(s->d, texture, duv, n, d, nullptr);
    PRT_HIP(hipDeviceSynchronize());
    PRT_HIP(hipMemcpy(rgb, d, n * 24, hipMemcpyDeviceToHost));
    return PRT_OK;
}

int prt_render_device(PrtScene* s, const PrtCamera* cam, const PrtRenderParams* p, void* d_rgb_f64, void* d_rgb_f32,
                      int count_work, void* stream) {
    int rc = require_uploaded(s, "prt_render_device");
    if (rc) return rc;
    if (!cam || !p) return fail(PRT_E_INVALID, "prt_render_device: null argument");
    if (cam->width < 1 || cam->height < 1) return fail(PRT_E_INVALID, "prt_render_device: bad image size");
    if (p->spp < 1) return fail(PRT_E_INVALID, "prt_render_device: spp must be >= 1");
    if (p->precision != PRT_PRECISION_F64 && p->precision != PRT_PRECISION_F32) return fail(PRT_E_INVALID, "prt_render_device: unsupported precision");
    const bool f32 = p->precision == PRT_PRECISION_F32;
    if (f32 && (rc = ensure_f32(s))) return rc;
    if (p->nranks < 1 || p->rank < 0 || p->rank >= p->nranks) return fail(PRT_E_INVALID, "prt_render_device: bad rank/nranks");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);

    DCamera C;
    prt::setup_camera(*cam, C);
    DRenderParams P;
    std::memset(&P, 0, sizeof(P));
    P.spp = p->spp;
    P.max_depth = p->max_depth;
    P.sample_lights = p->sample_lights ? 1 : 0;
    P.rr = p->russian_roulette;
    P.inv_rr = 1.0 / p->russian_roulette;
    // wave scheduling thresholds (developer overrides through the environment for sweeps)
    // measured optima at the BASELINE spp with 4-wide nodes (flat within 2 %): lean 28 / 48 / 20, others 20 / 40 / 12; the Phong
    // permutations at three waves: leaf batch 32 (round 4, two sweeps and a four-fold A/B on veach-mis: -0.75 %)
    P.keep = s->feat == 0 ? 28 : 20;
    P.leaf_batch = s->feat == 0 ? 48 : (s->feat & 2) ? 32 : 40; // (2 = Phong, as in s->feat above)
    P.inner_min = s->feat == 0 ? 20 : 12;
    P.cached_min = 24; // measured: veach-mis -1 %, the others flat
    if (const char* e = dev_env("PRT_TUNE_CACHED_MIN")) P.cached_min = std::max(1, std::atoi(e)); // (0 would keep a wave passing for ever)
    if (const char* e = dev_env("PRT_TUNE_KEEP")) P.keep = std::atoi(e);
    if (const char* e = dev_env("PRT_TUNE_LEAF_BATCH")) P.leaf_batch = std::atoi(e);
    if (const char* e = dev_env("PRT_TUNE_INNER_MIN")) P.inner_min = std::atoi(e);
    if (const char* e = dev_env("PRT_TUNE_SCRAMBLE")) P.scramble = std::atoi(e) ? 1 : 0; // experiment: incoherent pixel order (PRT_ITEMS_FROM_LIST stays internal to prt_render_samples)
    for (int c = 0; c < 3; ++c) P.background[c] = p->background[c];
    P.seed_key = prt::seed_key(p->seed); // the seed is hashed on its own, once per launch (prt_device.h, Rng)
    int tile = p->tile_size > 0 ? p->tile_size : 32;
    tile = std::max(8, (tile + 7) / 8 * 8);
    P.tile = tile;
    P.tiles_x = (C.width + tile - 1) / tile;
    P.tiles_y = (C.height + tile - 1) / tile;
    P.n_tiles = P.tiles_x * P.tiles_y;
    P.rank = p->rank;
    P.nranks = p->nranks;
    P.jitter = p->pixel_jitter ? 1 : 0;
    P.light_lds = s->light_lds;
    P.mat_lds = s->mat_lds;
    P.ltri_lds = s->ltri_lds;
    P.stack_depth = s->stack_depth;
    P.owned_tiles = P.n_tiles > P.rank ? (P.n_tiles - P.rank + P.nranks - 1) / P.nranks : 0;
    P.items_per_chunk = (uint64_t)P.owned_tiles * tile * tile;
    const bool count = count_work != 0;
    const int bpc = f32 ? s->blocks_per_cu32[count ? 1 : 0] : s->blocks_per_cu[count ? 1 : 0];
    const uint64_t lanes = (uint64_t)s->n_cu * bpc * PRT_BLOCK;
    // Work item = (pixel, chunk of samples), dealt chunk-major from PRT_ITEM_QUEUES counters (prt_types.h).
    //  * explicit sample_chunks: that many equal chunks;
    //  * auto: guided self-scheduling.  The launch ends when the LAST lane finishes, so no item may be
    //    handed out that can outlast the work still queued behind it.  Items of one chunk cover every owned
    //    pixel, pixels differ in cost by a factor `var` (a pixel looking into the box interior traces ~3x the
    //    rays of the average one), and `lanes` lanes drain the queue, so chunk j may hold at most
    //        s_j <= (pixels / lanes) / var * (samples in all later chunks)
    //    samples.  Built from the last chunk (1 sample) backwards this gives a geometric tail whose ratio
    //    depends on the share: x2.8 per chunk for a full 1024^2 frame, x1.2 for a 1/8 tile share — where
    //    a fixed halving tail left lanes finishing 32-sample items 5.8 ms after the queue ran dry (measured
    //    with per-wave timestamps: 58.3 ms launch, queue dry at 52.5 ms).  Sizes are capped at `body`
    //    samples; the per-item fetch is one wave-aggregated atomic.
    std::vector<int> sizes;
    const int spp = p->spp;
    int want = p->sample_chunks;
    if (const char* e = dev_env("PRT_TUNE_CHUNKS")) want = std::atoi(e);
    if (want > 0) {
        want = std::min(want, std::min(spp, PRT_MAX_CHUNKS));
        for (int c = 0; c < want; ++c) sizes.push_back((int)(((int64_t)(c + 1) * spp) / want - ((int64_t)c * spp) / want));
    } else if (P.items_per_chunk == 0) {
        sizes.push_back(spp);
    } else {
        // largest item: beyond ~100 samples the per-item costs are already amortised; a small tile share (fewer owned
        // pixels than resident lanes x 2) does 1 % better with 64 (53.1 -> 52.6 ms on a 1/8 share of the cornell frame)
        int body = P.items_per_chunk < 2 * lanes ? 64 : 128;
        if (const char* e = dev_env("PRT_TUNE_BODY")) body = std::max(1, std::atoi(e));
        body = std::max(body, (spp + PRT_MAX_CHUNKS / 2 - 1) / (PRT_MAX_CHUNKS / 2)); // very high spp: the body must fit in half the table
        double var = 4.0; // measured optimum with the multi-queue item dealing (3 before it: short items were fetch-bound)
        if (const char* e = dev_env("PRT_TUNE_VAR")) var = std::max(0.25, std::atof(e));
        double c = ((double)P.items_per_chunk / (double)lanes) / var;
        for (;;) {
            sizes.clear();
            int64_t sum = 0;
            while (sum < spp && (int)sizes.size() <= PRT_MAX_CHUNKS) {
                int64_t sz = (int64_t)std::floor(c * (double)sum);
                sz = std::max<int64_t>(1, std::min<int64_t>(sz, body));
                sz = std::min<int64_t>(sz, spp - sum);
                sizes.push_back((int)sz);
                sum += sz;
            }
            if ((int)sizes.size() <= PRT_MAX_CHUNKS) break;
            c *= 1.25; // a tiny share would need more chunks than the partial-sum table holds: steepen the tail
            if (c > 1e9) body *= 2; // (cannot happen with the body bound above; keeps the loop finite regardless)
        }
        std::reverse(sizes.begin(), sizes.end()); // largest chunks first
    }
    sizes.erase(std::remove(sizes.begin(), sizes.end(), 0), sizes.end());
    if (sizes.empty()) sizes.push_back(spp);
    int chunks = (int)sizes.size();
    P.chunk_begin[0] = 0;
    for (int c = 0; c < chunks; ++c) P.chunk_begin[c + 1] = P.chunk_begin[c] + sizes[c];
    P.chunks = chunks;
    P.n_items = P.items_per_chunk * (uint64_t)chunks;
    if (P.n_items >= 0xffffffffULL) return fail(PRT_E_LIMIT, "prt_render_device: more than 2^32 work items (pixels x sample chunks) in one launch");

    const size_t need = std::max<size_t>(P.n_items * 3, 3);
    hipError_t we;
    PrtScene::CallSlot& q = *s->next_slot(st, &we);
    PRT_HIP(we);
    if (need > q.partial_cap) {
        if (q.d_partial) (void)hipFree(q.d_partial); // hipFree waits for the device: nothing in flight reads it
        q.d_partial = nullptr;
        q.partial_cap = 0;
        PRT_HIP(hipMalloc(reinterpret_cast<void**>(&q.d_partial), need * sizeof(double)));
        q.partial_cap = need;
    }
    const size_t npx = (size_t)C.width * C.height * 3;
    void* dump_buf = nullptr;
    unsigned long long dump_cap = 0;
    std::string dump_path;
    PRT_HIP(hipMemsetAsync(q.d_ctr, 0, sizeof(DCounters), st));
    if (d_rgb_f64) PRT_HIP(hipMemsetAsync(d_rgb_f64, 0, npx * sizeof(double), st));
    if (d_rgb_f32) PRT_HIP(hipMemsetAsync(d_rgb_f32, 0, npx * sizeof(float), st));
    PRT_HIP(hipEventRecord(q.ev0, st));
    // maxDepth < 0: RayColor returns 0 before it traces anything (Camera.cpp:121) — the cleared framebuffer is the frame
    if (P.max_depth < 0) P.n_items = 0;
    if (P.n_items) {
        const uint64_t want = (P.n_items + PRT_BLOCK - 1) / PRT_BLOCK;
        const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)s->n_cu * bpc));
        if ((rc = set_slot_pointers(s, q, st, grid, nullptr, nullptr))) return rc;
        if (const char* e = dev_env("PRT_TUNE_DUMP_RAYS")) { // "<max rays>,<file>" (counting launches only)
            const std::string v(e);
            const size_t comma = v.find(',');
            if (count && comma != std::string::npos) {
                dump_cap = std::strtoull(v.substr(0, comma).c_str(), nullptr, 10);
                dump_path = v.substr(comma + 1);
                PRT_HIP(hipMalloc(&dump_buf, (size_t)dump_cap * sizeof(PrtRay)));
                const unsigned long long vals[3] = {(unsigned long long)(uintptr_t)dump_buf, dump_cap, 0ULL};
                PRT_HIP(hipMemcpyAsync(reinterpret_cast<char*>(q.d_ctr) + offsetof(DCounters, ray_dump), vals, sizeof(vals), hipMemcpyHostToDevice, st));
            }
        }
        if (f32) {
            // the same camera and parameters rounded to float (K5 below works from the fp64 originals: it only maps pixels)
            DCameraT<float> C32;
            conv_arr(C32.center, C.center, 3); conv_arr(C32.pixel00, C.pixel00, 3);
            conv_arr(C32.du, C.du, 3); conv_arr(C32.dv, C.dv, 3);
            C32.width = C.width; C32.height = C.height;
            DRenderParamsT<float> P32;
            std::memset(&P32, 0, sizeof(P32));
            P32.spp = P.spp; P32.max_depth = P.max_depth; P32.sample_lights = P.sample_lights; P32.chunks = P.chunks;
            P32.rr = (float)P.rr; P32.inv_rr = (float)P.inv_rr;
            conv_arr(P32.background, P.background, 3);
            P32.seed_key = P.seed_key;
            P32.tile = P.tile; P32.tiles_x = P.tiles_x; P32.tiles_y = P.tiles_y; P32.n_tiles = P.n_tiles;
            P32.rank = P.rank; P32.nranks = P.nranks; P32.owned_tiles = P.owned_tiles; P32.jitter = P.jitter;
            P32.keep = P.keep; P32.leaf_batch = P.leaf_batch; P32.inner_min = P.inner_min; P32.scramble = P.scramble; P32.cached_min = P.cached_min;
            P32.light_lds = s->light_lds32; P32.mat_lds = s->mat_lds32; P32.ltri_lds = s->ltri_lds32;
            P32.stack_depth = s->stack_depth32;
            P32.items_per_chunk = P.items_per_chunk; P32.n_items = P.n_items;
            std::memcpy(P32.chunk_begin, P.chunk_begin, sizeof(P.chunk_begin));
            prt32::launch_render(s->d32, C32, P32, q.d_partial, q.d_ctr, count, s->feat, grid, st);
        } else {
            prt::launch_render(s->d_k3, C, P, q.d_partial, q.d_ctr, count, s->feat, grid, st);
        }
        PRT_HIP(hipGetLastError());
    }
    PRT_HIP(hipEventRecord(q.ev1, st));
    if (dump_buf) { // developer experiment: write K3's ray stream to the file (synchronous)
        PRT_HIP(hipStreamSynchronize(st));
        DCounters h;
        PRT_HIP(hipMemcpy(&h, q.d_ctr, sizeof(h), hipMemcpyDeviceToHost));
        const size_t nd = (size_t)std::min<unsigned long long>(h.ray_dump_n, dump_cap);
        std::vector<PrtRay> rays(nd);
        PRT_HIP(hipMemcpy(rays.data(), dump_buf, nd * sizeof(PrtRay), hipMemcpyDeviceToHost));
        (void)hipFree(dump_buf);
        if (FILE* f = std::fopen(dump_path.c_str(), "wb")) {
            std::fwrite(rays.data(), sizeof(PrtRay), nd, f);
            std::fclose(f);
        }
        std::fprintf(stderr, "[prt] dumped %zu of %llu rays to %s\n", nd, h.ray_dump_n, dump_path.c_str());
    }
    if (P.n_items) {
        prt::launch_finalize(C, P, q.d_partial, static_cast<double*>(d_rgb_f64), static_cast<float*>(d_rgb_f32), st);
        PRT_HIP(hipGetLastError());
    }
    PRT_HIP(hipEventRecord(q.done, st));
    q.timed = true;
    q.counted = count;
    q.samples = 0;
    if (P.n_items) { // pixels of this rank's tiles that lie inside the image, times spp
        uint64_t px = 0;
        for (int k = P.rank; k < P.n_tiles; k += P.nranks) {
            const int ty = k / P.tiles_x, kx = k - ty * P.tiles_x, tx = (kx + 3 * ty) % P.tiles_x; // prt_device.h, owned_to_pixel
            px += (uint64_t)std::max(0, std::min(tile, C.width - tx * tile)) * (uint64_t)std::max(0, std::min(tile, C.height - ty * tile));
        }
        q.samples = px * (uint64_t)spp;
    }
    return PRT_OK;
}

int prt_render(PrtScene* s, const PrtCamera* cam, const PrtRenderParams* p, double* rgb_f64, float* rgb_f32) {
    int rc = require_uploaded(s, "prt_render");
    if (rc) return rc;
    if (!cam || !p) return fail(PRT_E_INVALID, "prt_render: null argument");
    if (cam->width < 1 || cam->height < 1) return fail(PRT_E_INVALID, "prt_render: bad image size");
    const size_t npx = (size_t)cam->width * cam->height * 3;
    void *d64 = nullptr, *d32 = nullptr;
    if (rgb_f64) PRT_HIP(hipMalloc(&d64, npx * sizeof(double)));
    if (rgb_f32) {
        if (hipMalloc(&d32, npx * sizeof(float)) != hipSuccess) {
            if (d64) (void)hipFree(d64);
            return fail(PRT_E_OOM, "prt_render: hipMalloc failed");
        }
    }
    rc = prt_render_device(s, cam, p, d64, d32, 0, nullptr);
    if (rc == PRT_OK) {
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) rc = fail(PRT_E_HIP, std::string("prt_render: ") + hipGetErrorString(e));
    }
    if (rc == PRT_OK && rgb_f64 && hipMemcpy(rgb_f64, d64, npx * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(PRT_E_HIP, "prt_render: D2H copy failed");
    if (rc == PRT_OK && rgb_f32 && hipMemcpy(rgb_f32, d32, npx * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(PRT_E_HIP, "prt_render: D2H copy failed");
    if (d64) (void)hipFree(d64);
    if (d32) (void)hipFree(d32);
    return rc;
}

// Test hook (prt.h): single camera samples through K3.  One work item per (listed pixel, sample): the sample chunks of a
// launch are the samples themselves (at most PRT_MAX_CHUNKS per launch), spp = 1 so that nothing is scaled, and the item's
// partial sum IS the sample's RayColor.
int prt_render_samples(PrtScene* s, const PrtCamera* cam, const PrtRenderParams* p, const int32_t* pixel_xy, size_t n_pixels,
                       int32_t sample_begin, int32_t sample_count, double* radiance, int32_t* trace) {
    int rc = require_uploaded(s, "prt_render_samples");
    if (rc) return rc;
    if (!cam || !p || (n_pixels && (!pixel_xy || !radiance))) return fail(PRT_E_INVALID, "prt_render_samples: null argument");
    if (cam->width < 1 || cam->height < 1) return fail(PRT_E_INVALID, "prt_render_samples: bad image size");
    if (p->precision != PRT_PRECISION_F64) return fail(PRT_E_INVALID, "prt_render_samples: fp64 only");
    if (sample_begin < 0 || sample_count < 0) return fail(PRT_E_INVALID, "prt_render_samples: bad sample range");
    if (n_pixels == 0 || sample_count == 0) return PRT_OK;
    if (n_pixels * (size_t)PRT_MAX_CHUNKS >= 0xffffffffULL) return fail(PRT_E_LIMIT, "prt_render_samples: too many pixels");
    std::vector<int32_t> pix(n_pixels);
    for (size_t k = 0; k < n_pixels; ++k) {
        const int32_t i = pixel_xy[2 * k], j = pixel_xy[2 * k + 1];
        if (i < 0 || j < 0 || i >= cam->width || j >= cam->height) return fail(PRT_E_INVALID, "prt_render_samples: pixel outside the image");
        pix[k] = j * cam->width + i;
    }
    DCamera C;
    prt::setup_camera(*cam, C);
    DRenderParams P;
    std::memset(&P, 0, sizeof(P));
    P.spp = 1;
    P.max_depth = p->max_depth;
    P.sample_lights = p->sample_lights ? 1 : 0;
    P.rr = p->russian_roulette;
    P.inv_rr = 1.0 / p->russian_roulette;
    P.keep = s->feat == 0 ? 28 : 20;
    P.leaf_batch = s->feat == 0 ? 48 : (s->feat & 2) ? 32 : 40; // (2 = Phong, as in s->feat above)
    P.inner_min = s->feat == 0 ? 20 : 12;
    P.scramble = PRT_ITEMS_FROM_LIST;
    P.cached_min = 65;
    for (int c = 0; c < 3; ++c) P.background[c] = p->background[c];
    P.seed_key = prt::seed_key(p->seed);
    P.tile = 8;
    P.tiles_x = P.tiles_y = P.n_tiles = 1;
    P.rank = 0;
    P.nranks = 1;
    P.owned_tiles = 1;
    P.jitter = p->pixel_jitter ? 1 : 0;
    P.light_lds = s->light_lds;
    P.mat_lds = s->mat_lds;
    P.ltri_lds = s->ltri_lds;
    P.stack_depth = s->stack_depth;
    P.items_per_chunk = n_pixels;
    const bool count = trace != nullptr;
    const int bpc = s->blocks_per_cu[count ? 1 : 0];
    int32_t *d_pix = nullptr, *d_trace = nullptr;
    double* d_part = nullptr;
    const size_t per_launch = n_pixels * (size_t)std::min<int>(sample_count, PRT_MAX_CHUNKS);
    PRT_HIP(hipMalloc(reinterpret_cast<void**>(&d_pix), n_pixels * sizeof(int32_t)));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_part), per_launch * 3 * sizeof(double));
    if (e == hipSuccess && trace) e = hipMalloc(reinterpret_cast<void**>(&d_trace), per_launch * PRT_TRACE_WORDS * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(d_pix, pix.data(), n_pixels * sizeof(int32_t), hipMemcpyHostToDevice);
    std::vector<double> part(per_launch * 3);
    std::vector<int32_t> tr(trace ? per_launch * PRT_TRACE_WORDS : 0);
    hipError_t we = hipSuccess;
    for (int32_t s0 = 0; e == hipSuccess && s0 < sample_count; s0 += PRT_MAX_CHUNKS) {
        const int chunks = std::min<int>(PRT_MAX_CHUNKS, sample_count - s0);
        for (int c = 0; c <= chunks; ++c) P.chunk_begin[c] = sample_begin + s0 + c;
        P.chunks = chunks;
        P.n_items = P.items_per_chunk * (uint64_t)chunks;
        PrtScene::CallSlot& q = *s->next_slot(nullptr, &we);
        if ((e = we) != hipSuccess) break;
        if ((e = hipMemsetAsync(q.d_ctr, 0, sizeof(DCounters), nullptr)) != hipSuccess) break;
        const uint64_t want = (P.n_items + PRT_BLOCK - 1) / PRT_BLOCK;
        const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(want, (uint64_t)s->n_cu * bpc));
        if (set_slot_pointers(s, q, nullptr, grid, d_pix, d_trace) != PRT_OK) { e = hipErrorUnknown; break; }
        if (d_trace && (e = hipMemset(d_trace, 0, (size_t)P.n_items * PRT_TRACE_WORDS * sizeof(int32_t))) != hipSuccess) break;
        (void)hipEventRecord(q.ev0, nullptr);
        if (P.max_depth >= 0) prt::launch_render(s->d_k3, C, P, d_part, q.d_ctr, count, s->feat, grid, nullptr);
        else if ((e = hipMemset(d_part, 0, (size_t)P.n_items * 3 * sizeof(double))) != hipSuccess) break;
        if ((e = hipGetLastError()) != hipSuccess) break;
        (void)hipEventRecord(q.ev1, nullptr);
        (void)hipEventRecord(q.done, nullptr);
        q.timed = true;
        q.counted = count;
        q.samples = P.max_depth >= 0 ? P.n_items : 0;
        if ((e = hipDeviceSynchronize()) != hipSuccess) break;
        if ((e = hipMemcpy(part.data(), d_part, (size_t)P.n_items * 3 * sizeof(double), hipMemcpyDeviceToHost)) != hipSuccess) break;
        if (trace && (e = hipMemcpy(tr.data(), d_trace, (size_t)P.n_items * PRT_TRACE_WORDS * sizeof(int32_t), hipMemcpyDeviceToHost)) != hipSuccess) break;
        for (int c = 0; c < chunks; ++c)
            for (size_t k = 0; k < n_pixels; ++k) { // item = chunk * n_pixels + k  ->  out[k][s0 + c]
                const size_t item = (size_t)c * n_pixels + k, o = k * (size_t)sample_count + (size_t)(s0 + c);
                std::memcpy(radiance + o * 3, part.data() + item * 3, 3 * sizeof(double));
                if (trace) std::memcpy(trace + o * PRT_TRACE_WORDS, tr.data() + item * PRT_TRACE_WORDS, PRT_TRACE_WORDS * sizeof(int32_t));
            }
    }
    (void)hipFree(d_pix);
    if (d_part) (void)hipFree(d_part);
    if (d_trace) (void)hipFree(d_trace);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? PRT_E_OOM : PRT_E_HIP, std::string("prt_render_samples: ") + hipGetErrorString(e));
    return PRT_OK;
}

namespace {
// One RCCL communicator set per list of devices, created on first use and kept for the life of the process
// (ncclCommInitAll takes hundreds of milliseconds; a frame takes tens).
struct CommSet {
    std::vector<ncclComm_t> comms;
};
std::mutex g_comm_mutex;
std::map<std::vector<int>, CommSet> g_comms;
} // namespace

namespace {
// Restores the caller's current HIP device on every way out of a call that visits several devices.
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};
void drop_comms_locked(std::map<std::vector<int>, CommSet>::iterator it, bool abort) {
    for (ncclComm_t c : it->second.comms)
        if (c) (void)(abort ? ncclCommAbort(c) : ncclCommDestroy(c));
    g_comms.erase(it);
}

int render_multi_impl(PrtScene* const* scenes, int n, const PrtCamera* cam, const PrtRenderParams* p, float* rgb_f32) {
    std::vector<int> devs(n);
    for (int r = 0; r < n; ++r) {
        if (!scenes[r]) return fail(PRT_E_INVALID, "prt_render_multi: null scene");
        if (scenes[r]->device < 0) return fail(PRT_E_NO_DEVICE, "prt_render_multi: a scene is not uploaded to a HIP device (no CPU path exists)");
        for (int q = 0; q < r; ++q)
            if (scenes[q] == scenes[r]) return fail(PRT_E_INVALID, "prt_render_multi: the same scene handle twice (one handle per tile share)");
        devs[r] = scenes[r]->device;
    }
    bool all_same = true, all_distinct = true;
    for (int r = 0; r < n; ++r)
        for (int q = 0; q < r; ++q) {
            if (devs[q] == devs[r]) all_distinct = false;
            else all_same = false;
        }
    if (n > 1 && !all_same && !all_distinct)
        return fail(PRT_E_INVALID, "prt_render_multi: scenes must sit on pairwise different devices (RCCL reduce) or all on one device");
    const size_t npx = (size_t)cam->width * cam->height * 3;
    // every share renders its tiles into its device's zeroed full-size fp32 framebuffer
    for (int r = 0; r < n; ++r) {
        PrtScene* s = scenes[r];
        PRT_HIP(hipSetDevice(s->device));
        if (s->multi_fb_cap < npx) {
            if (s->multi_fb) (void)hipFree(s->multi_fb);
            s->multi_fb = nullptr;
            s->multi_fb_cap = 0;
            PRT_HIP(hipMalloc(reinterpret_cast<void**>(&s->multi_fb), npx * sizeof(float)));
            s->multi_fb_cap = npx;
        }
        PrtRenderParams pr = *p;
        pr.tile_size = n > 1 ? 16 : p->tile_size;
        pr.rank = r;
        pr.nranks = n;
        const int rc = prt_render_device(s, cam, &pr, nullptr, s->multi_fb, 0, nullptr);
        if (rc != PRT_OK) return rc;
    }
    // test hooks (PRT_DEV_HOOKS builds only).  PRT_TEST_FORCE_RCCL: a single scene goes through the RCCL branch as well (a
    // communicator of one rank) — the most of that branch a one-GPU box can execute: library, communicator, the grouped
    // reduce on the device buffer, the stream order.  PRT_TEST_FAIL_NCCL=init|reduce: that step reports a failure.
    const bool force_rccl = n == 1 && dev_env("PRT_TEST_FORCE_RCCL") && std::atoi(dev_env("PRT_TEST_FORCE_RCCL"));
    const char* inject = dev_env("PRT_TEST_FAIL_NCCL");
    if (n > 1 && all_same) {
        // tile shares of one device (replicas; a rehearsal of the multi-GPU path on one GPU): summed where they are
        PRT_HIP(hipSetDevice(devs[0]));
        for (int r = 1; r < n; ++r) prt::launch_add_f32(scenes[0]->multi_fb, scenes[r]->multi_fb, npx, nullptr);
        PRT_HIP(hipGetLastError());
    } else if (n > 1 || force_rccl) {
        // ONE collective: reduce(sum) of the fp32 framebuffers to the first device over RCCL (xGMI between the GPUs of a
        // node).  Tiles are disjoint, so every element is x + 0 + ... + 0: the reduce is exact.  The communicator set of a
        // device list is created once and kept (prt_shutdown destroys them); the mutex is held for the whole collective,
        // so two host threads cannot interleave grouped calls on the same communicators.
        std::lock_guard<std::mutex> lock(g_comm_mutex);
        auto it = g_comms.find(devs);
        if (it == g_comms.end()) {
            CommSet fresh;
            fresh.comms.assign(n, nullptr);
            ncclResult_t nr = (inject && !std::strcmp(inject, "init")) ? ncclSystemError : ncclCommInitAll(fresh.comms.data(), n, devs.data());
            if (nr != ncclSuccess) {
                for (ncclComm_t c : fresh.comms) // whatever a failed initialisation left behind
                    if (c) (void)ncclCommAbort(c);
                return fail(PRT_E_HIP, std::string("prt_render_multi: ncclCommInitAll failed: ") + ncclGetErrorString(nr) +
                                           " (no host-side fallback exists: the frame is not assembled)");
            }
            it = g_comms.emplace(devs, std::move(fresh)).first;
        }
        CommSet& cs = it->second;
        // Every error between ncclGroupStart and ncclGroupEnd is COLLECTED: the group is always closed before the call fails.
        std::string what;
        ncclResult_t nr = ncclGroupStart();
        if (nr != ncclSuccess) return fail(PRT_E_HIP, std::string("prt_render_multi: ncclGroupStart failed: ") + ncclGetErrorString(nr));
        for (int r = 0; r < n && nr == ncclSuccess; ++r) {
            const hipError_t he = hipSetDevice(devs[r]);
            if (he != hipSuccess) {
                what = std::string("hipSetDevice: ") + hipGetErrorString(he);
                nr = ncclUnhandledCudaError;
                break;
            }
            nr = (inject && !std::strcmp(inject, "reduce")) ? ncclInternalError
                 : ncclReduce(scenes[r]->multi_fb, scenes[r]->multi_fb, npx, ncclFloat, ncclSum, 0, cs.comms[r], nullptr);
        }
        const ncclResult_t ne = ncclGroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) {
            drop_comms_locked(it, true); // a communicator that failed a collective is not used again
            return fail(PRT_E_HIP, std::string("prt_render_multi: ncclReduce failed: ") + ncclGetErrorString(nr) + (what.empty() ? "" : " (" + what + ")"));
        }
        for (int r = 1; r < n; ++r) { // the root's copy below only waits for the root's stream
            PRT_HIP(hipSetDevice(devs[r]));
            PRT_HIP(hipDeviceSynchronize());
        }
    }
    PRT_HIP(hipSetDevice(devs[0]));
    PRT_HIP(hipDeviceSynchronize());
    PRT_HIP(hipMemcpy(rgb_f32, scenes[0]->multi_fb, npx * sizeof(float), hipMemcpyDeviceToHost));
    return PRT_OK;
}
} // namespace

// Camera::Render over several GPUs of this process (prt.h).  Single host thread: every device's launches are
// asynchronous; the reduce is one grouped RCCL call.  The caller's current device is restored on every way out.
// EXPERIMENTAL for n > 1 on different devices: that branch has only ever run with a one-rank communicator (rounds 1-4
// had one-GPU boxes); the result is an fp32 framebuffer (the reduce's element type), not the fp64 one of prt_render.
int prt_render_multi(PrtScene* const* scenes, int n, const PrtCamera* cam, const PrtRenderParams* p, float* rgb_f32) {
    if (!scenes || n < 1 || !cam || !p || !rgb_f32) return fail(PRT_E_INVALID, "prt_render_multi: null argument");
    if (cam->width < 1 || cam->height < 1) return fail(PRT_E_INVALID, "prt_render_multi: bad image size");
    DeviceGuard guard;
    return render_multi_impl(scenes, n, cam, p, rgb_f32);
}

// Releases what the library keeps for the whole process: the cached RCCL communicators of prt_render_multi.  Scenes are
// not touched (prt_scene_destroy).  Call it before the process exits when prt_render_multi ran over several devices; safe
// to call any number of times, and prt_render_multi creates communicators again when it needs them.
void prt_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_comm_mutex);
    DeviceGuard guard;
    while (!g_comms.empty()) drop_comms_locked(g_comms.begin(), false);
}

int prt_get_counters(PrtScene* s, PrtCounters* out) {
    if (!s || !out) return fail(PRT_E_INVALID, "prt_get_counters: null argument");
    PrtCounters c = s->last;
    c.bvh_nodes = s->bvh_info.n_nodes; // host- and device-built trees alike (bvh.nodes is empty for the latter)
    c.bvh_depth = s->bvh_info.depth;
    PrtScene::CallSlot& q = s->slots[s->cur];
    if (s->device >= 0 && q.timed) {
        PRT_HIP(hipSetDevice(s->device));
        PRT_HIP(hipEventSynchronize(q.ev1));
        float ms = 0.f;
        PRT_HIP(hipEventElapsedTime(&ms, q.ev0, q.ev1));
        DCounters h;
        PRT_HIP(hipMemcpy(&h, q.d_ctr, sizeof(h), hipMemcpyDeviceToHost));
        c.rays_closest = h.rays_closest;
        c.rays_shadow = h.rays_shadow;
        c.node_fetches = h.node_fetches;
        c.tri_tests = h.tri_tests;
        c.samples = q.samples;
        c.inner_rounds = h.inner_rounds;
        c.leaf_rounds = h.leaf_rounds;
        c.refills = h.refills;
        c.tri_full = h.tri_full;
        c.kernel_ms = ms;
    }
    s->last = c;
    *out = c;
    return PRT_OK;
}

int prt_tonemap_srgb8(PrtScene* s, const void* d_rgb_f32, int width, int height, void* d_rgb_u8, void* stream) {
    int rc = require_uploaded(s, "prt_tonemap_srgb8");
    if (rc) return rc;
    if (!d_rgb_f32 || !d_rgb_u8 || width < 1 || height < 1) return fail(PRT_E_INVALID, "prt_tonemap_srgb8: bad argument");
    prt::launch_tonemap(static_cast<const float*>(d_rgb_f32), (size_t)width * height * 3, static_cast<uint8_t*>(d_rgb_u8),
                        reinterpret_cast<hipStream_t>(stream));
    PRT_HIP(hipGetLastError());
    return PRT_OK;
}

} // extern "C"

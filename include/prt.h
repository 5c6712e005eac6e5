/*
 * prt.h — C ABI of libprt_hip.so, the MI355X (gfx950) path-tracing hot path.
 *
 * This is the drop-in boundary for the reference's per-pixel / per-sample hot path
 *     Camera::Render -> RayColor -> BVHNode::Hit -> Triangle::Hit / AABB::Hit
 * (reference: Source/Camera.cpp:21-204, Source/BVH.cpp:51-61, Source/Triangle.cpp:54-83,
 *  Source/AABB.cpp:38-64).  The reference has no FFI layer of its own; the
 * entry points below are what a binding for that path would call:
 *
 *   prt_scene_create      replaces the object graph main.cpp:36-45 builds
 *                         (Mesh/Triangle ctor precompute Source/Triangle.cpp:11-53,
 *                          two-level BVHNode build Source/BVH.cpp:6-49, lights list main.cpp:40-45)
 *   prt_trace_closest     replaces world.Hit(ray, Interval(tmin,tmax), record)
 *                         (Source/HittableList.h:26-39 -> Source/BVH.cpp:51-61)
 *   prt_render            replaces Camera::Render(world, lights)      (Source/Camera.cpp:21-73)
 *   prt_render_device     same, framebuffer left in device memory for an RCCL reduce
 *   prt_render_multi      same over several GPUs of this process: tiles + one RCCL reduce of the fp32 framebuffer
 *                         (replaces the std::thread row bands of Source/Camera.cpp:46-71)
 *   prt_sample_lights     replaces lights.Sample(origin, record, pdf) (Source/HittableList.h:44-59,
 *                          Source/BVH.cpp:62-67,86-100, Source/Triangle.cpp:84-93) — test hook
 *   prt_get_counters      rays / node fetches / triangle tests / kernel ms of the last call
 *
 * Conventions: every function returns 0 on success or a negative PRT_E_* code and never throws;
 * prt_last_error() returns a thread-local message for the last failure.  All input buffers are
 * owned by the caller and may be freed as soon as the call returns.  Handles are opaque.  One
 * host thread per device; calls on different handles are independent.  No CPU fallback exists:
 * without a HIP device every compute entry point fails with PRT_E_NO_DEVICE.
 */
#ifndef PRT_H
#define PRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRT_ABI_VERSION 5

/* error codes */
#define PRT_OK 0
#define PRT_E_INVALID (-1)    /* bad argument / inconsistent scene description / non-finite vertex coordinate */
#define PRT_E_NO_DEVICE (-2)  /* no HIP device visible */
#define PRT_E_HIP (-3)        /* a HIP runtime call failed (message has the HIP error string) */
#define PRT_E_OOM (-4)        /* host or device allocation failed */
#define PRT_E_LIMIT (-5)      /* scene exceeds a compiled-in limit (BVH depth, leaf encoding) */

/* Material kinds — reference enum MaterialType, Source/Material.h:48-51 */
#define PRT_MAT_LAMBERTIAN 0    /* Source/Material.h:101-155 */
#define PRT_MAT_PHONG 1         /* PhoneReflectance, Source/Material.h:172-330 */
#define PRT_MAT_MIRROR 2        /* PerfectMirror, Source/Material.h:332-366 */
#define PRT_MAT_COOKTORRANCE 3  /* Source/Material.h:368-521 */
#define PRT_MAT_DIFFUSE_LIGHT 4 /* Source/Material.h:157-170 */
#define PRT_MAT_DEBUG 5         /* Source/Material.h:523-535 (emits its albedo) */
#define PRT_MAT_EMPTY 6         /* Source/Material.h:537-540 */

typedef struct PrtMaterial {
    int32_t type;      /* PRT_MAT_* */
    int32_t texture;   /* index into PrtSceneDesc.textures for the Kd map, or -1 (SolidColor) */
    double kd[3];      /* Lambertian albedo / Phong Kd / Debug albedo */
    double ks[3];      /* Phong Ks (ignored when texture >= 0: reference stores mapKd in both, Material.h:178-181) */
    double ns;         /* Phong exponent */
    double emission[3];/* DiffuseLight radiance (XML <light radiance>, Source/Model.cpp:332-360) */
    double eta[3];     /* CookTorrance conductor eta */
    double k[3];       /* CookTorrance conductor k */
    double alpha_x, alpha_y; /* CookTorrance roughness */
} PrtMaterial;

/* 8-bit interleaved texels exactly as stbi_load returns them (Source/Texture.cpp:10-21). */
typedef struct PrtTexture {
    int32_t width, height, channels, reserved;
    const uint8_t* data; /* width*height*channels bytes; NULL => reference's "missing" colour (0,1,1) */
} PrtTexture;

/*
 * Scene = list of meshes; a mesh = contiguous triangle range + one material
 * (reference: Mesh, Source/Triangle.h:36-43; one material per shape, Source/Model.cpp:118).
 * Triangle order inside a mesh and mesh order are significant: they are the input order of the
 * reference's std::sort-based light-tree build, which fixes the NEE light CDF order.
 */
typedef struct PrtSceneDesc {
    uint64_t n_tris;
    const double* vertices;  /* [n_tris][3 verts][xyz] */
    const double* normals;   /* [n_tris][3][xyz] vertex normals (degenerate-face fallback only) or NULL */
    const double* texcoords; /* [n_tris][3][uv] or NULL (all zero) */
    uint32_t n_meshes;
    uint32_t n_materials;
    const uint64_t* mesh_first_tri; /* [n_meshes+1], ascending, last == n_tris */
    const int32_t* mesh_material;   /* [n_meshes] index into materials */
    const PrtMaterial* materials;
    uint32_t n_textures;
    uint32_t flags;          /* PRT_SCENE_* bits; 0 = defaults */
    const PrtTexture* textures;
    /* The `lights` argument of Camera::Render(world, lights) (Source/Camera.h:27): indices of the meshes that were
     * added to the lights list, in that order.  NULL = what main.cpp:40-45 builds — every mesh whose material
     * HasEmission(), in mesh order.  A non-NULL pointer with n_light_meshes == 0 is an empty list (no NEE). */
    const int32_t* light_meshes;
    uint32_t n_light_meshes;
    uint32_t reserved;
} PrtSceneDesc;

/* Build the traversal BVH on the GPU at prt_scene_upload time instead of on the host at create time
 * (reference: BVHNode constructors, Source/BVH.cpp:7-48, always CPU).  Results do not depend on the
 * builder; the host builder stays the default because its trees traverse a few percent faster. */
#define PRT_SCENE_DEVICE_BVH 1u

/* Public camera fields of the reference, Source/Camera.h:14-24. */
typedef struct PrtCamera {
    int32_t width, height;
    double fovy; /* degrees */
    double eye[3], look_at[3], up[3];
} PrtCamera;

#define PRT_PRECISION_F64 0 /* reference arithmetic (glm::dvec3 everywhere) */
/* fp32 fast mode: the same kernels with every real number a float (48-byte triangle records, hardware rcp / rsq / sqrt).
 * Same random streams, same tree; results agree with PRT_PRECISION_F64 within the second tolerance tier (hits
 * |dt|/t <= 1e-5, images statistically: knife-edge branches differ), not to 1e-9.  The float tables are derived from
 * the resident fp64 ones on the first call that asks for them (that call is synchronous). */
#define PRT_PRECISION_F32 1

typedef struct PrtRenderParams {
    int32_t spp;           /* Camera::samplesPerPixel */
    int32_t max_depth;     /* Camera::maxDepth (maxDepth+1 path vertices, Camera.cpp:121) */
    double russian_roulette; /* Camera::russianRoulette */
    int32_t sample_lights; /* Camera::bSampleLights */
    int32_t precision;     /* PRT_PRECISION_* */
    double background[3];  /* Camera::background */
    uint64_t seed;         /* per-sample RNG key = (seed, j*W+i, s) */
    int32_t tile_size;     /* multi-GPU tile edge in pixels (0 => 32) */
    int32_t rank, nranks;  /* this device renders tiles k with k % nranks == rank; others stay 0 */
    int32_t sample_chunks; /* 0 => auto; partial sums per pixel are combined in fixed order */
    int32_t pixel_jitter;  /* 0 = reference behaviour (pixel centre).  1 = the anti-aliasing the reference has
                              commented out (Camera.cpp:110-111): SampleSquare() offset, drawn per sample as
                              the first two numbers of the sample's stream (offset.y first, offset.x second) */
    int32_t reserved;      /* must be 0 */
} PrtRenderParams;

/* One ray of a batch: world.Hit(Ray(o,d), Interval(tmin,tmax)). */
typedef struct PrtRay {
    double o[3];
    double tmin;
    double d[3];
    double tmax;
} PrtRay;

/* HitRecord subset that identifies the hit (Source/Hittable.h:17-28). */
typedef struct PrtHit {
    double t;      /* HitRecord::time; +inf on miss */
    double alpha;  /* barycentric of v1 (Triangle.cpp:69) */
    double beta;   /* barycentric of v2 (Triangle.cpp:70) */
    int32_t prim;  /* triangle index in PrtSceneDesc order, -1 on miss */
    int32_t front; /* HitRecord::bFrontFace */
} PrtHit;

/* lights.Sample() result (test hook). */
typedef struct PrtLightSample {
    double position[3];
    double normal[3]; /* face-forwarded against (p - origin), Triangle.cpp:89-90 */
    double pdf;       /* 1 / total light area */
    int32_t prim;
    int32_t front;
} PrtLightSample;

typedef struct PrtCounters {
    uint64_t rays_closest;  /* camera + continuation traversals executed (the camera ray of a pixel is traced once per work item, not per sample) */
    uint64_t rays_shadow;   /* NEE visibility traversals */
    uint64_t node_fetches;  /* BVH node records read (counting runs only; PrtBvhInfo.node_bytes each) */
    uint64_t tri_tests;     /* triangle plane / interval tests = the first 32 bytes (n, D) of a 96-byte record (counting runs only) */
    uint64_t samples;       /* camera samples of the call (pixels rendered x spp) */
    double kernel_ms;       /* hipEvent time of the dominant kernel of the last call */
    uint64_t bvh_nodes;     /* static: nodes in the flattened tree */
    uint64_t bvh_depth;     /* static: max depth */
    uint64_t inner_rounds;  /* counting runs: wave-level node-visit rounds (64 lanes each) */
    uint64_t leaf_rounds;   /* counting runs: wave-level leaf rounds */
    uint64_t refills;       /* counting runs: wave-level shade/refill passes */
    uint64_t tri_full;      /* counting runs: tests that passed the interval check and fetched the 64 bytes of edge functions */
} PrtCounters;

/* Which builder made the traversal BVH and what it cost. */
typedef struct PrtBvhInfo {
    uint64_t n_nodes;
    uint32_t depth;
    uint32_t built_on_device;
    double build_ms;  /* host builder: wall time inside prt_scene_create; device builder: HIP-event time */
    double sort_ms, tree_ms, split_ms; /* device builder phases: Morton sort / box segment tree / SAH levels */
    uint32_t node_bytes; /* size of one node record in HBM */
    uint32_t width;      /* children per node */
    uint32_t tri_bytes;  /* payload of one intersection record: 32 (plane: n, D) + the part read after the interval test */
    uint32_t tri_stride; /* bytes between records in HBM once uploaded (0 before): tri_bytes, or 128 for scenes that stream from HBM */
    /* image textures as resident on the device (0 before upload; ImageTexture::Value, Source/Texture.cpp:22-71).  A scene's
     * textures are stored as bilinear footprints (per texel cell the four taps of a lookup: 128 bytes per texel, one line per
     * lookup) while ALL of them together stay within 256 MiB; a scene beyond that keeps plain texel arrays (24 bytes per
     * texel).  The fp32 fast mode adds a float copy of half the size on first use. */
    uint64_t texture_bytes;           /* fp64 bytes of all texel arrays */
    uint64_t texture_footprint_bytes; /* ... of which footprint records */
    uint32_t texture_layouts;         /* 0: no textures; 1: footprints; 2: plain texel arrays (one layout per scene) */
    /* how the fp64 render kernel (K3) of this scene is launched (0 before upload) */
    uint32_t render_blocks_per_cu;    /* resident 256-thread blocks per CU (= waves per SIMD) of the production instantiation */
    uint32_t render_blocks_wanted;    /* ... the register allocation of the scene's material permutation leaves room for */
    uint32_t lds_materials, lds_light_nodes, lds_light_tris; /* shading tables K3 stages in LDS (0: read from global memory) */
    uint32_t stack_need;              /* traversal stack entries this tree can need (<= 40, the builders' bound) */
    uint32_t reserved_;
} PrtBvhInfo;

typedef struct PrtScene PrtScene;

int prt_abi_version(void);
/* 1 when this build reads the developer / test environment hooks (PRT_TUNE_*, PRT_TEST_*; -DPRT_DEV_HOOKS=1 builds only —
 * libprt_hip_dev.so of the test suite).  The shipped libprt_hip.so returns 0 and reads no environment variable. */
int prt_dev_hooks(void);
const char* prt_last_error(void);
int prt_device_count(int* n);

int prt_scene_create(const PrtSceneDesc* desc, PrtScene** out);
void prt_scene_destroy(PrtScene* scene);
/* Upload nodes / triangles / materials / textures / light tree to `device` (the tree was built in prt_scene_create on
 * the host, or is built here on the GPU with PRT_SCENE_DEVICE_BVH).  All or nothing: on failure nothing stays resident
 * and the scene is back in the not-uploaded state. */
int prt_scene_upload(PrtScene* scene, int device);

/* New vertex positions ([n_tris][3][xyz], same layout as PrtSceneDesc.vertices; normals may be NULL) for a
 * scene whose topology, materials and texture coordinates stay as created: re-runs the Triangle constructor
 * precompute and the light tree; on an uploaded scene the BVH is REBUILT on the GPU (PRT_SCENE_DEVICE_BVH
 * path: 4 ms per 126k triangles, 17 ms per 8M).  Limitation: there is no topology-preserving refit (the
 * reference has none either — it rebuilds, Source/BVH.cpp:7-48); a caller that moves geometry every frame pays
 * the rebuild plus a re-upload of the triangle records each time. */
int prt_scene_update_vertices(PrtScene* scene, const double* vertices, const double* normals);

/* Number of light triangles and their order in the reference's area-CDF descent (BVH.cpp:86-100). */
int prt_scene_bvh_info(const PrtScene* scene, PrtBvhInfo* out);

int prt_scene_light_count(const PrtScene* scene, uint64_t* n);
int prt_scene_light_order(const PrtScene* scene, int32_t* prims, uint64_t cap);

/* K1: closest hit for a batch of host rays; with count_work != 0 the counting instantiation runs. */
int prt_trace_closest(PrtScene* scene, const PrtRay* rays, size_t n, PrtHit* hits, int count_work);
/* K1 on device-resident buffers (d_rays/d_hits are device pointers); stream may be NULL. */
int prt_trace_closest_device(PrtScene* scene, const void* d_rays, size_t n, void* d_hits,
                             int count_work, void* hip_stream);
/* Same with a PRT_PRECISION_* choice (rays and hits stay fp64 records at the boundary). */
int prt_trace_closest_device_prec(PrtScene* scene, const void* d_rays, size_t n, void* d_hits,
                                  int count_work, int precision, void* hip_stream);

/* K4 + K1: the same batch traced in a locality order.  A pre-pass sorts (cell of the origin in the scene's box, octant of
 * the direction) keys on the device and K1 takes the rays in that order, so that the lanes of a wave walk the same part
 * of the tree — for scenes whose BVH and triangles exceed the caches (several million triangles) and batches of
 * incoherent rays, where every node visit is otherwise a line of its own from HBM.  hits[i] still answers rays[i], bit
 * for bit what prt_trace_closest_device_prec returns; the time reported by prt_get_counters covers keys + sort + trace.
 * Needs 16 bytes of scratch per ray (kept by the scene between calls); n < 2^32.  On cache-resident scenes the sort costs
 * more than it returns. */
int prt_trace_closest_sorted_device(PrtScene* scene, const void* d_rays, size_t n, void* d_hits,
                                    int count_work, int precision, void* hip_stream);

/* NEE point selection for (pixel, sample) keys 0..n-1 of `seed` from given origins (test hook). */
int prt_sample_lights(PrtScene* scene, const double* origins, size_t n, uint64_t seed,
                      PrtLightSample* out);

/*
 * Test hooks for the material arithmetic K3 shades with (the device functions themselves, on caller-supplied
 * directions).  Item i draws from the stream keyed (seed, i, 0).  `material` / `texture` index the scene's tables.
 *   prt_material_eval     Material::Eval(wi, ctx{wo, uv}), wi / wo LOCAL (z = shading normal):
 *                         Lambertian Material.h:128-130, PhoneReflectance :227-248 (draws one number), CookTorrance :474-496
 *   prt_material_scatter  Material::Scatter(Ray(0, rd_i), record{normal, tangent, uv}) (Material.h:131-151,263-285,
 *                         344-363,497-516): scattered direction in WORLD space, attenuation = f cos / pdf, ok = its return value
 *   prt_texture_value     ImageTexture::Value(u, v) (Texture.cpp:22-49)
 * uv may be NULL (all zero).
 */
int prt_material_eval(PrtScene* scene, int32_t material, size_t n, const double* wi, const double* wo, const double* uv,
                      uint64_t seed, double* f);
int prt_material_scatter(PrtScene* scene, int32_t material, size_t n, const double* rd, const double* normal,
                         const double* tangent, const double* uv, uint64_t seed, double* wi_world, double* attenuation,
                         int32_t* ok);
int prt_texture_value(PrtScene* scene, int32_t texture, size_t n, const double* uv, double* rgb);

/*
 * K3+K5: render one frame.  rgb_f64 / rgb_f32 are W*H*3 row-major host buffers (either may be
 * NULL).  Pixels of tiles owned by other ranks are written as 0 so a sum over ranks is exact.
 */
int prt_render(PrtScene* scene, const PrtCamera* cam, const PrtRenderParams* params,
               double* rgb_f64, float* rgb_f32);
/* Same, outputs are device pointers on the scene's device; asynchronous on hip_stream.  A scene keeps two sets of
 * per-call state (counters, partial sums, events): two calls — render or trace — may be in flight at once on different
 * streams; a third one first waits (on its stream) for the call that last used its set.  A PrtScene is not thread-safe:
 * issue its calls from one host thread. */
int prt_render_device(PrtScene* scene, const PrtCamera* cam, const PrtRenderParams* params,
                      void* d_rgb_f64, void* d_rgb_f32, int count_work, void* hip_stream);

/*
 * Camera::Render over several GPUs of one process — the reference's only parallel split is the thread fan-out over row
 * bands inside Camera::Render (Source/Camera.cpp:46-71); here the frame is cut into 16x16 tiles dealt over the scenes:
 * scenes[r] is the SAME scene description uploaded to a different device each (prt_scene_upload).  Every device renders
 * its tiles into a zeroed full-size fp32 framebuffer, ONE RCCL reduce(sum, float) to scenes[0]'s device assembles the
 * frame (disjoint tiles: x + 0 + ... + 0 — an exact reduce; the single-GPU fp32 image bit for bit when sample_chunks is
 * explicit, else up to the fp64 rounding of a share's own chunking, ~1e-15) and one copy brings it to rgb_f32
 * (W*H*3 floats).  n == 1 is prt_render's fp32 output.  All scenes on ONE device (tile-share replicas) are summed on
 * that device without a collective; any other mix is refused.  If the RCCL communicator cannot be created, or the
 * reduce fails, the call fails (PRT_E_HIP): there is no host-side sum to fall back to; the scenes stay usable and the
 * caller's current HIP device is restored on every way out.  Communicators are cached per device list until prt_shutdown.
 * EXPERIMENTAL for n > 1 on different devices: that branch has so far only run with a communicator of one rank (no
 * multi-GPU box was available to rounds 1-4).  The result has fp32 precision (the element type of the reduce), where
 * prt_render's rgb_f64 is the fp64 frame.
 */
int prt_render_multi(PrtScene* const* scenes, int n, const PrtCamera* cam, const PrtRenderParams* params, float* rgb_f32);
/* Releases process-wide state: the RCCL communicators prt_render_multi cached (ncclCommDestroy).  Scenes are untouched.
 * Call before exit after multi-device renders; may be called repeatedly, and prt_render_multi re-creates what it needs. */
void prt_shutdown(void);

/*
 * Test hook: RayColor of single camera samples through K3 itself (the production instantiation of k_render when `trace`
 * is NULL, its counting instantiation otherwise) — what orc_render_samples is for the oracle.  Sample s of pixel
 * (pixel_xy[2k], pixel_xy[2k+1]) draws from the stream keyed (params->seed, j*W+i, s) like in a frame, whatever params->spp
 * says; radiance[k][s - sample_begin][3] is that sample's RayColor (not divided by spp).  params->precision must be
 * PRT_PRECISION_F64; rank / tile / chunk fields are ignored.
 * trace (optional) [n_pixels][sample_count][PRT_TRACE_WORDS]: the path's signature — word 0 = path vertices visited;
 * then per vertex v (at most PRT_TRACE_VERTS): word 1+2v = triangle hit (PrtSceneDesc order, -1 = miss), word 2+2v = flags:
 *   PRT_TRACE_NEE       the light sample passed n.wi > 0 and faces the shading point (Camera.cpp:153-154): a shadow ray was traced
 *   PRT_TRACE_VISIBLE   ... and it reached the light: direct light was added (Camera.cpp:155-172)
 *   PRT_TRACE_ROULETTE  RandomDouble() < russianRoulette (Camera.cpp:180)
 *   PRT_TRACE_SCATTER   Material::Scatter returned true (Camera.cpp:182)
 */
#define PRT_TRACE_WORDS 64
#define PRT_TRACE_VERTS 31
#define PRT_TRACE_NEE 1
#define PRT_TRACE_VISIBLE 2
#define PRT_TRACE_ROULETTE 4
#define PRT_TRACE_SCATTER 8
int prt_render_samples(PrtScene* scene, const PrtCamera* cam, const PrtRenderParams* params, const int32_t* pixel_xy,
                       size_t n_pixels, int32_t sample_begin, int32_t sample_count, double* radiance, int32_t* trace);

int prt_get_counters(PrtScene* scene, PrtCounters* out);

/* K5 "next" row: NaN scrub + linear->sRGB + clamp -> 8-bit RGB (Camera.cpp:206-221,279-301). */
int prt_tonemap_srgb8(PrtScene* scene, const void* d_rgb_f32, int width, int height,
                      void* d_rgb_u8, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* PRT_H */

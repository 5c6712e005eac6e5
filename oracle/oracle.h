/*
 * oracle.h — C ABI of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under pooraytracer_amd/ or include/ may include, link or
 * load this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / reported baseline — never as the thing shipped or measured as the product.
 *
 * PARITY UNPINNED: the reference (Zoz4/Pooraytracer) has no tests, golden vectors or fixtures for
 * this path, its third-party headers (glm, spdlog, stb, tinyxml2, tinyobjloader) are
 * absent so it is unbuildable here without stand-in headers (which the build rules forbid), and it
 * is C++ so it cannot be imported.  This oracle is therefore a line-by-line CPU restatement of the
 * reference algorithm (each function cites the reference file:line it follows), pinned only by
 * hand-derived known-answer tests (tests/test_oracle.py), closed forms / normalisation integrals /
 * chi-square tests of the material arithmetic from the literature (tests/test_materials.py), a second
 * independent restatement of Camera::RayColor in Python that must agree per sample
 * (tests/test_oracle_crosscheck.py) and its own committed fixtures in tests/golden/ — none of which
 * is a reference output.
 *
 * The struct layouts deliberately equal those of include/prt.h so a test can hand the same
 * buffers to both libraries; they are re-declared here so the oracle has no product dependency.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcMaterial {
    int32_t type;
    int32_t texture;
    double kd[3];
    double ks[3];
    double ns;
    double emission[3];
    double eta[3];
    double k[3];
    double alpha_x, alpha_y;
} OrcMaterial;

typedef struct OrcTexture {
    int32_t width, height, channels, reserved;
    const uint8_t* data;
} OrcTexture;

typedef struct OrcSceneDesc {
    uint64_t n_tris;
    const double* vertices;
    const double* normals;
    const double* texcoords;
    uint32_t n_meshes;
    uint32_t n_materials;
    const uint64_t* mesh_first_tri;
    const int32_t* mesh_material;
    const OrcMaterial* materials;
    uint32_t n_textures;
    uint32_t reserved;
    const OrcTexture* textures;
    const int32_t* light_meshes; /* the `lights` list as mesh indices, NULL = every emissive mesh in mesh order (main.cpp:40-45) */
    uint32_t n_light_meshes;
    uint32_t reserved2;
} OrcSceneDesc;

typedef struct OrcCamera {
    int32_t width, height;
    double fovy;
    double eye[3], look_at[3], up[3];
} OrcCamera;

typedef struct OrcRenderParams {
    int32_t spp;
    int32_t max_depth;
    double russian_roulette;
    int32_t sample_lights;
    int32_t precision;
    double background[3];
    uint64_t seed;
    int32_t tile_size;
    int32_t rank, nranks;
    int32_t sample_chunks;
    int32_t pixel_jitter; /* 0 = reference (pixel centre); 1 = SampleSquare offset per sample (Camera.cpp:110-111, disabled upstream) */
    int32_t reserved;
} OrcRenderParams;

typedef struct OrcRay {
    double o[3];
    double tmin;
    double d[3];
    double tmax;
} OrcRay;

typedef struct OrcHit {
    double t;
    double alpha;
    double beta;
    int32_t prim;
    int32_t front;
} OrcHit;

typedef struct OrcLightSample {
    double position[3];
    double normal[3];
    double pdf;
    int32_t prim;
    int32_t front;
} OrcLightSample;

typedef struct OrcCounters {
    uint64_t rays_closest; /* unique camera + continuation traversals (peek re-trace not counted) */
    uint64_t rays_shadow;
    uint64_t hit_calls;    /* world.Hit calls actually executed */
    uint64_t samples;
} OrcCounters;

typedef struct OrcScene OrcScene;

OrcScene* orc_scene_create(const OrcSceneDesc* desc);
void orc_scene_destroy(OrcScene*);
uint64_t orc_light_count(const OrcScene*);
void orc_light_order(const OrcScene*, int32_t* prims);

/* world.Hit(ray, Interval(tmin,tmax), record) for each ray. */
void orc_trace_closest(const OrcScene*, const OrcRay* rays, size_t n, OrcHit* hits);
/* lights.Sample(origin_i) with the stream keyed (seed, i, 0). */
void orc_sample_lights(const OrcScene*, const double* origins, size_t n, uint64_t seed,
                       OrcLightSample* out);
/* first `n` uniforms of the stream keyed (seed, pixel, sample). */
void orc_rng_stream(uint64_t seed, uint64_t pixel, uint64_t sample, size_t n, double* out);

/*
 * Camera::Render with per-sample keyed RNG.  rows [y0,y1) only (others untouched); nthreads row-
 * interleaved workers; reuse_peek != 0 hands the peek hit (Camera.cpp:187) to the recursive call
 * instead of re-tracing the identical ray (output identical, asserted by tests).
 */
void orc_render(const OrcScene*, const OrcCamera*, const OrcRenderParams*, double* rgb_f64,
                int y0, int y1, int nthreads, int reuse_peek, OrcCounters* counters);
/* per-sample radiance of selected pixels: out[n_pixels][spp][3]. */
void orc_render_samples(const OrcScene*, const OrcCamera*, const OrcRenderParams*,
                        const int32_t* pixel_xy, size_t n_pixels, double* out);
/* Same for samples [sample_begin, sample_begin + sample_count) — out[n_pixels][sample_count][3] — and, when `trace` is not
 * NULL, the signature of every path: trace[n_pixels][sample_count][ORC_TRACE_WORDS], word 0 = path vertices visited, then
 * per vertex v (RayColor recursion level v): word 1+2v = triangle hit (description order, -1 = miss), word 2+2v = flags.
 * Same layout and meaning as the product's prt_render_samples (include/prt.h). */
#define ORC_TRACE_WORDS 64
#define ORC_TRACE_VERTS 31
#define ORC_TRACE_NEE 1       /* dot(n, wi) > 0 && light sample front-facing (Camera.cpp:153-154) */
#define ORC_TRACE_VISIBLE 2   /* ... and visible: direct light added (Camera.cpp:155-172) */
#define ORC_TRACE_ROULETTE 4  /* RandomDouble() < russianRoulette (Camera.cpp:180) */
#define ORC_TRACE_SCATTER 8   /* Material::Scatter returned true (Camera.cpp:182) */
void orc_render_samples_trace(const OrcScene*, const OrcCamera*, const OrcRenderParams*, const int32_t* pixel_xy,
                              size_t n_pixels, int32_t sample_begin, int32_t sample_count, double* out, int32_t* trace);
/* Material / texture hooks for the known-answer tests.  Directions are LOCAL (z = shading normal) for eval and the
 * CookTorrance terms, WORLD for scatter; item i uses the stream keyed (seed, i, 0).
 *   orc_material_eval      Material::Eval(wi, ctx{wo, uv})                              -> f[n][3]
 *   orc_material_scatter   Material::Scatter(Ray(0, rd_i), record{normal, tangent, uv}) -> scattered direction (world),
 *                          attenuation = f cos / pdf, ok = Scatter's return value
 *   orc_texture_value      ImageTexture::Value(u, v)
 *   orc_cooktorrance_terms {D(wm), Lambda(w), G1(w), D(w, wm)} and Fresnel(|w . wm|) per channel
 *   orc_frame              Material::WorldToLocal / LocalToWorld */
void orc_material_eval(const OrcScene*, int32_t material, size_t n, const double* wi, const double* wo, const double* uv,
                       uint64_t seed, double* f);
void orc_material_scatter(const OrcScene*, int32_t material, size_t n, const double* rd, const double* normal,
                          const double* tangent, const double* uv, uint64_t seed, double* wi_world, double* attenuation,
                          int32_t* ok);
void orc_texture_value(const OrcScene*, int32_t texture, size_t n, const double* uv, double* rgb);
void orc_cooktorrance_terms(const OrcScene*, int32_t material, size_t n, const double* w, const double* wm, double* out4,
                            double* fresnel3);
void orc_frame(const double* normal, const double* tangent, size_t n, const double* in, int to_local, double* out);
/* camera rays (origin, unnormalised direction) for all pixels: out[H][W][6] (Camera.cpp:75-117). */
void orc_camera_rays(const OrcCamera*, double* out);

#ifdef __cplusplus
}
#endif
#endif

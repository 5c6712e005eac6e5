// prt_kernels.hip — gfx950 kernels of the path-tracing hot path and their launchers.
//
//   K1 k_trace_closest   closest hit for a ray batch               (world.Hit, BVH.cpp:51-61)
//   K3 k_render          persistent-wavefront path tracer          (Camera::Render/RayColor, Camera.cpp:21-204)
//   K5 k_finalize        ordered sum of per-chunk partial sums -> f64 / f32 framebuffer
//      k_sample_lights   lights.Sample test hook                   (BVH.cpp:62-67,86-100)
//      k_tonemap         NaN scrub + sRGB + clamp -> u8            (Camera.cpp:206-221,279-301)
//
// K3 design (MI355X: 256 CUs x 4 SIMD, wave64, 160 KB LDS/CU, per-XCD L2):
//   * persistent workgroups (grid = CUs x resident blocks); every LANE owns one work item =
//     (pixel, sample chunk) and pulls the next one from one of 16 global atomic counters when it finishes, so
//     short paths (light / background pixels) never idle a wave for long — lane-level regeneration
//     instead of a per-bounce compaction pass;
//   * traversal is resumable per lane (Trav::step = one node visit or one leaf).  A wave steps all
//     its traversing lanes together and leaves the stepping loop as soon as only PRT_K3_KEEP of them
//     are still busy (ballot + popcount); the finished lanes consume their hit (shade / NEE set-up /
//     Russian roulette / Scatter / next sample / next item) and start their next ray — continuation
//     or shadow, whichever that path needs — while the others keep their stack and resume.  Lanes
//     never wait for the slowest ray of the wave, and there is a single traversal call site;
//     shadow rays are any-hit rays over [0.001, dist - 0.001] (what Camera.cpp:150-155's test amounts to);
//   * the camera ray of a pixel is traced once per work item; every sample starts from its parked hit;
//   * traversal stack in LDS, lane-strided (conflict-free), PRT_STACK_DEPTH entries per lane;
//   * results are deterministic: per-sample keyed RNG, per-item partial sums combined in a fixed
//     order by K5 (no float atomics on the framebuffer).
#include <hip/hip_runtime.h>

#include "../../include/prt.h"
#include "prt_device.h"

// This file is compiled twice: as it stands (fp64, the reference's arithmetic: namespace prt, every kernel) and through
// prt_kernels_f32.hip with PRT_REAL = float (the fp32 fast mode: namespace prt32, K1 and K3 only — K5, the tone map and
// the test hooks work on fp64 buffers in either mode and exist once).
#ifndef PRT_F32_TU
#define PRT_F32_TU 0
#endif
#if PRT_F32_TU
#define PRT_NS prt32
#else
#define PRT_NS prt
#endif
#define PRT_DYN_STACK PRT_F32_TU // K3's traversal stacks in dynamic LDS, sized per launch from what the scene's tree can need (DRenderParams::stack_depth): the fp32 kernels, which have the registers for a fourth block per CU; the fp64 kernels are register-limited to three and keep static PRT_STACK_DEPTH-entry stacks (a run-time depth cost them 1.2 %)

// minimum resident waves per SIMD the register allocator must leave room for in K3
#ifndef PRT_RENDER_WAVES
#define PRT_RENDER_WAVES 2
#endif
#ifndef PRT_RENDER_WAVES_LEAN
#define PRT_RENDER_WAVES_LEAN 3 // the lean material permutation fits one more wave per SIMD
#endif
#ifndef PRT_RENDER_WAVES_TEX
#define PRT_RENDER_WAVES_TEX 3  // Lambertian / mirror / light + image textures (bathroom2-class scenes)
#endif
#ifndef PRT_RENDER_WAVES_PHONG
#define PRT_RENDER_WAVES_PHONG 3 // PhoneReflectance without textures (veach-mis-class scenes): fits since round 3 (168 registers); pays since round 4 (the light tables freed the LDS the third block needs)
#endif
#ifndef PRT_F32_WAVES
#define PRT_F32_WAVES 3 // fp32 fast mode: resident waves per SIMD of every K3 permutation
#endif
constexpr int render_waves(int feat_with_extra) {
    const int feat = feat_with_extra & PRT_FEAT_ALL; // (PRT_FEAT_EXTRA is not a material feature)
    if (PRT_F32_TU) return PRT_F32_WAVES;
    return feat == 0 ? PRT_RENDER_WAVES_LEAN
         : feat == PRT_FEAT_TEX ? PRT_RENDER_WAVES_TEX
         : feat == PRT_FEAT_PHONG ? PRT_RENDER_WAVES_PHONG
         : PRT_RENDER_WAVES;
}
// The stepping loop of a wave runs while MORE than this many lanes are still traversing; below it the
// finished lanes are handed new rays (K1) / shaded and re-armed (K3).
#ifndef PRT_K1_KEEP
#define PRT_K1_KEEP 48
#endif
#ifndef PRT_K3_KEEP
#define PRT_K3_KEEP 24
#endif
#ifndef PRT_K1_WAVES
#define PRT_K1_WAVES 4 // resident waves per SIMD the K1 register allocation leaves room for
#endif
#ifndef PRT_K1_CHUNK
#define PRT_K1_CHUNK 1024
#endif

namespace {

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

#ifndef PRT_K3_PROFILE
#define PRT_K3_PROFILE 0 // developer diagnostic (COUNT instantiation): shader-clock cycles per section of the wave loop, folded into the counters
#endif
#if PRT_K3_PROFILE
#define PROF_MARK(k) do { if (COUNT) { const unsigned long long t_ = __builtin_readcyclecounter(); prof_[k] += t_ - prof_t_; prof_t_ = t_; } } while (0)
#else
#define PROF_MARK(k) do { } while (0)
#endif
// ------------------------------------------------------------------------------------------- K1
// Persistent waves; every lane pulls its next ray from a global counter the moment its traversal
// ends.  The stepping loop is left (and the finished lanes refilled) once no more than
// PRT_K1_KEEP lanes are still traversing.
template <bool COUNT, bool PAD>
__global__ __launch_bounds__(PRT_BLOCK, PRT_K1_WAVES) void k_trace_closest(DScene S, const PrtRay* __restrict__ rays, size_t n,
                                                             PrtHit* __restrict__ hits, DCounters* ctr,
                                                             const uint32_t* __restrict__ perm) { // K4's order (ray_sort.hip) or null
    __shared__ uint32_t s_stack[PRT_BLOCK / 64][PRT_STACK_DEPTH][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* stk = &s_stack[wave][0][lane];
    WorkCount wc{0, 0, 0, 0, 0};
    uint32_t nrays = 0;
    Trav<PAD> tr;
    tr.init(S, mk3(0, 0, 0), mk3(0, 0, 1), RL(0.0), RL(0.0));
    tr.hit.alpha = tr.hit.beta = RL(0.0);
    tr.active = false;
    bool have = false;
    size_t my = 0;
    // wave-local ray pool [pool_next, pool_end): refilled PRT_K1_CHUNK rays at a time by ONE atomic per
    // wave (a single global counter saturates at ~90 dequeues/us on this chip)
    unsigned long long pool_next = 0, pool_end = 0;
    bool exhausted = false; // wave-uniform
    for (;;) {
        if (!tr.active) {
            if (have) {
                PrtHit out;
                if (tr.hit.tri >= 0) {
                    const DTri* T = tri_at<PAD>(S, (uint32_t)tr.hit.tri);
                    const d3 nrm = mk3(T->n[0], T->n[1], T->n[2]);
                    out.t = tr.hit.t;
                    out.alpha = tr.hit.alpha;
                    out.beta = tr.hit.beta;
                    out.prim = S.shade[tr.hit.tri].prim;
                    out.front = dot(tr.d, nrm) < RL(0.) ? 1 : 0; // HitRecord::SetFaceNormal, Hittable.cpp:8-13
                } else {
                    out.t = PRT_INF;
                    out.alpha = out.beta = 0;
                    out.prim = -1;
                    out.front = 0;
                }
                hits[my] = out;
                have = false;
            }
        }
        {
            const unsigned long long need = __ballot(!tr.active);
            if (need != 0ULL && !exhausted) {
                if (pool_next >= pool_end) {
                    unsigned long long base = 0;
                    if (lane == (int)__builtin_ctzll(need)) base = atomicAdd(&ctr->next_item, (unsigned long long)PRT_K1_CHUNK);
                    base = __shfl(base, (int)__builtin_ctzll(need), 64);
                    pool_next = base;
                    pool_end = base + PRT_K1_CHUNK < (unsigned long long)n ? base + PRT_K1_CHUNK : (unsigned long long)n;
                    if (base >= (unsigned long long)n) exhausted = true;
                }
                if (!exhausted && !tr.active) {
                    const unsigned long long below = need & ((1ULL << lane) - 1ULL);
                    const unsigned long long idx = pool_next + (unsigned long long)__popcll(below);
                    if (idx < pool_end) {
                        const size_t ri = perm ? (size_t)perm[idx] : (size_t)idx; // the idx-th ray of the sorted order
                        const double4* rp = reinterpret_cast<const double4*>(rays + ri);
                        const double4 r0 = rp[0], r1 = rp[1];
                        tr.init(S, mk3((real)r0.x, (real)r0.y, (real)r0.z), mk3((real)r1.x, (real)r1.y, (real)r1.z), (real)r0.w, (real)r1.w); // PrtRay is fp64 at the ABI in either mode
                        my = ri;
                        have = true;
                        nrays++;
                    }
                }
                if (!exhausted) {
                    const unsigned long long taken = (unsigned long long)__popcll(need);
                    pool_next = pool_next + taken < pool_end ? pool_next + taken : pool_end;
                }
            }
        }
        if (__ballot(tr.active || have) == 0ULL && exhausted) break;
        do {
            tr.template round<COUNT>(S, stk, wc, PRT_LEAF_BATCH, PRT_INNER_MIN, tr.tmin, false);
        } while (wave_count(tr.active) > PRT_K1_KEEP);
    }
    unsigned long long a = wave_sum((unsigned long long)nrays);
    unsigned long long b = wave_sum((unsigned long long)wc.nodes);
    unsigned long long c = wave_sum((unsigned long long)wc.tris);
    unsigned long long f = wave_sum((unsigned long long)wc.tris_full);
    if (lane == 0) {
        atomicAdd(&ctr->rays_closest, a);
        if (COUNT) {
            atomicAdd(&ctr->node_fetches, b);
            atomicAdd(&ctr->tri_tests, c);
            atomicAdd(&ctr->tri_full, f);
        }
    }
}

// ------------------------------------------------------------------------------------------- K3
enum : int { ST_FETCH = 0, ST_NEW_SAMPLE = 1, ST_CLOSEST = 2, ST_SHADOW = 3, ST_DONE = 4, ST_PRIMARY = 5, ST_CACHED = 6 };
// The camera ray of a pixel is the same for every sample (Camera.cpp:53-57: GetRay once per pixel, no jitter): K3 traces
// it ONCE per work item (ST_PRIMARY) and parks the ray's direction and its hit — t, triangle, barycentrics — in LDS,
// lane-strided; every sample of the item starts from that hit (ST_CACHED) instead of re-tracing the identical ray.

// Shading context of a hit, rebuilt from (incoming ray, HitInfo): HitRecord of Triangle::Hit
// (Triangle.cpp:76-80,111) — position = ray(t), face-forwarded normal, tangent, uv.
struct ShadeCtx {
    Frame f;
    d2 uv;
    int32_t material;
};
// `rd` = the direction the hit was reached along, (alpha, beta, tri) = the hit (read in place: no HitInfo copy)
template <int FEAT, bool PAD>
PRT_DEV ShadeCtx make_ctx(const DScene& S, d3 rd, real alpha, real beta, int32_t tri) {
    ShadeCtx c;
    const DTriShade* sh = S.shade + tri;
    const DTri* T = tri_at<PAD>(S, (uint32_t)tri);
    const d3 gn = mk3(T->n[0], T->n[1], T->n[2]);
    const bool front = dot(rd, gn) < RL(0.);
    c.f.n = front ? gn : -gn;
    c.f.t = ld3(sh->tangent);
    if (FEAT & PRT_FEAT_TEX) { // texture coordinates are only read by image-textured materials
        const real w0 = RL(1.) - alpha - beta;
        c.uv.x = w0 * sh->uv0[0] + alpha * sh->uv1[0] + beta * sh->uv2[0];
        c.uv.y = w0 * sh->uv0[1] + alpha * sh->uv1[1] + beta * sh->uv2[1];
    } else {
        c.uv.x = c.uv.y = RL(0.0);
    }
    c.material = sh->material;
    return c;
}

// A real number parked in / fetched from a lane's LDS column (`base` = the lane's slot of word 0, words PRT_BLOCK apart)
template <int PARK_STRIDE>
PRT_DEV void park_real(uint32_t* base, int word, real v) {
    if (PRT_F32) {
        base[word * PARK_STRIDE] = __float_as_uint((float)v);
    } else {
        const unsigned long long b = (unsigned long long)__double_as_longlong((double)v);
        base[word * PARK_STRIDE] = (uint32_t)b;
        base[(word + 1) * PARK_STRIDE] = (uint32_t)(b >> 32);
    }
}
template <int PARK_STRIDE>
PRT_DEV real unpark_real(const uint32_t* base, int word) {
    if (PRT_F32) return (real)__uint_as_float(base[word * PARK_STRIDE]);
    return (real)__longlong_as_double((long long)(((unsigned long long)base[(word + 1) * PARK_STRIDE] << 32) | base[word * PARK_STRIDE]));
}
#define PRT_RW ((int)(sizeof(real) / 4)) // dwords per real
// a lane's parked camera ray and primary hit: t | triangle | direction[3] | (textured permutations) alpha, beta
#define PARK_T 0
#define PARK_TRI PRT_RW
#define PARK_DIR (PRT_RW + 1)
// Textured permutations do not park the direction: their LDS is spoken for (a deep tree's 40-entry stacks plus t, triangle
// and barycentrics plus the material table fill a third of a CU), so they recompute it per sample from the camera, read on
// demand (measured on bathroom2: parked in global memory +4.5 %, parked in LDS at the price of the 32-entry tree +2.5 %).
#define PARK_DIR_GLOBAL(feat) (((feat) & PRT_FEAT_TEX) != 0)
#define PARK_AB_AT(feat) (PARK_DIR_GLOBAL(feat) ? PRT_RW + 1 : 4 * PRT_RW + 1)
#define PARK_WORDS(feat) (((feat) & PRT_FEAT_TEX) ? 3 * PRT_RW + 1 : 4 * PRT_RW + 1)

// K3's arguments are ONE struct, so that the kernel can also reach them through the kernarg segment pointer: what the
// hot loop needs (scene tables, thresholds, roulette, seed) is used as plain arguments — the compiler loads those into
// scalar registers once — while everything only a work-item fetch or a miss needs (the camera, tile geometry, the chunk
// table, the background, the output pointer: ~50 scalar registers) is read ON DEMAND through cold_args(), a pointer to the
// same bytes that the compiler cannot see through: s_load at the point of use instead of values kept (and, beyond ~100
// of them, spilled to VGPR lanes: 116 v_readlane per pass before this) across the whole traversal loop.
template <typename R> // (the scalar type shows in the kernel's name: profiles tell the fp64 launch from the fp32 one by it)
struct RenderArgsT {
    DSceneT<R> S;
    DCameraT<R> C;
    DRenderParamsT<R> P;
    double* partial;
    DCounters* ctr;
};
typedef RenderArgsT<real> RenderArgs;
typedef const RenderArgs __attribute__((address_space(4)))* ColdRenderArgs;
PRT_DEV ColdRenderArgs cold_args() {
    ColdRenderArgs q = (ColdRenderArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(q)); // opaque from here on: loads through q stay where they are written
    return q;
}

template <bool COUNT, int FEAT, bool LLDS, bool PAD>
__global__ __launch_bounds__(PRT_BLOCK, render_waves(FEAT)) void k_render(RenderArgs A) {
    const DScene& S = A.S;
    const DRenderParams& P = A.P; // hot fields only: see RenderArgs
    DCounters* const ctr = A.ctr;
    __shared__ uint32_t s_qoff[PRT_BLOCK / 64];
    __shared__ unsigned long long s_rays[PRT_BLOCK / 64];
    __shared__ real s_center[4]; // Camera::center, the origin of every camera ray
    constexpr int NPARK = PRT_BLOCK;
    __shared__ uint32_t s_park[PARK_WORDS(FEAT)][NPARK]; // per lane: the work item's camera ray and its hit (ST_PRIMARY), read once per sample
#define park (&s_park[0][threadIdx.x])
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // Dynamic LDS, sized by the host: the four waves' traversal stacks (P.stack_depth entries per lane, lane-strided),
    // then the shading tables.  The depth is a launch parameter: what the scene's tree can need, not the builders' bound of
    // PRT_STACK_DEPTH — LDS left over decides how many blocks a CU holds (the fp32 kernels have the registers for a fourth
    // block when a tree needs at most 32 entries) and leaves room for the parked camera rays.
    extern __shared__ __align__(16) unsigned char s_dyn_all[];
#if PRT_DYN_STACK
    uint32_t* stk = reinterpret_cast<uint32_t*>(s_dyn_all) + (size_t)(wave * P.stack_depth) * 64 + lane;
    unsigned char* s_dyn = s_dyn_all + (size_t)(PRT_BLOCK / 64) * P.stack_depth * 64 * sizeof(uint32_t);
#else
    __shared__ uint32_t s_stack[PRT_BLOCK / 64][PRT_STACK_DEPTH][64];
    uint32_t* stk = &s_stack[wave][0][lane];
    unsigned char* s_dyn = s_dyn_all;
#endif
    if (lane == 0) {
        s_qoff[wave] = 0;
        s_rays[wave] = 0ULL;
    }
    if (threadIdx.x < 3) s_center[threadIdx.x] = A.C.center[threadIdx.x];
    if (!LLDS) __syncthreads(); // (the LLDS kernels synchronise below, after staging their tables)
    // light tree in LDS (see sample_lights)
    const DLightNode* lds_lights = reinterpret_cast<const DLightNode*>(s_dyn);
    // ... followed by the whole material table (the host only selects LLDS when it fits): its fields are read
    // several times per path vertex, and every one of those reads is otherwise a texture-addresser instruction
    const DMaterial* lds_mats = reinterpret_cast<const DMaterial*>(s_dyn + (size_t)P.light_lds * sizeof(DLightNode));
    const DLightTri* lds_ltris = reinterpret_cast<const DLightTri*>(s_dyn + (size_t)P.light_lds * sizeof(DLightNode) + (size_t)P.mat_lds * sizeof(DMaterial));
    if (LLDS) { // every thread of the block gets here before any divergence
        uint4* dst = reinterpret_cast<uint4*>(s_dyn);
        const uint4* src = reinterpret_cast<const uint4*>(S.light_nodes);
        for (int i = threadIdx.x; i < P.light_lds; i += PRT_BLOCK) dst[i] = src[i];
        dst += P.light_lds;
        src = reinterpret_cast<const uint4*>(S.materials);
        const int nm = P.mat_lds * (int)(sizeof(DMaterial) / 16);
        for (int i = threadIdx.x; i < nm; i += PRT_BLOCK) dst[i] = src[i];
        dst += nm;
        src = reinterpret_cast<const uint4*>(S.light_tris);
        const int nl = P.ltri_lds * (int)(sizeof(DLightTri) / 16);
        for (int i = threadIdx.x; i < nl; i += PRT_BLOCK) dst[i] = src[i];
        __syncthreads();
    }
#define MATERIAL(i) (LLDS ? lds_mats[(i)] : S.materials[(i)])
    d3 pst_[2]; // acc (this item's sum of sample radiance / spp), beta (path throughput)
#define PST_LD(k) (pst_[(k) / 3])
#define PST_ST(k, v) (pst_[(k) / 3] = (v))
#define S_ACC 0
#define S_BETA 3
// A radiance term x met by the current path (emission, background, NEE): colorAttachment[m] += RayColor(...) *
// pixelSamplesScale (Camera.cpp:56) with RayColor unrolled into sum_k beta_k * x_k; each term is scaled and
// added on the spot, so no per-sample radiance has to live in registers across traversals.
#define ADD_RADIANCE(x) PST_ST(S_ACC, PST_LD(S_ACC) + (PST_LD(S_BETA) * (x)) * inv_spp)
// prt_render_samples with a trace buffer (counting instantiation, one sample per work item): the path's signature,
// vertex v = max_depth - depth (prt.h, PRT_TRACE_*)
#define TRACING (COUNT && P.scramble == PRT_ITEMS_FROM_LIST && ctr->trace != nullptr)
#define TRACE_VERTEX(prim_)                                                     \
    do {                                                                        \
        if (TRACING) {                                                          \
            const int v_ = P.max_depth - depth;                                 \
            if (v_ < PRT_TRACE_VERTS) {                                         \
                int32_t* tw_ = ctr->trace + (size_t)item * PRT_TRACE_WORDS;     \
                tw_[0] = v_ + 1;                                                \
                tw_[1 + 2 * v_] = (prim_);                                      \
                tw_[2 + 2 * v_] = 0;                                            \
            }                                                                   \
        }                                                                       \
    } while (0)
#define TRACE_FLAG(bit_)                                                        \
    do {                                                                        \
        if (TRACING) {                                                          \
            const int v_ = P.max_depth - depth;                                 \
            if (v_ < PRT_TRACE_VERTS) ctr->trace[(size_t)item * PRT_TRACE_WORDS + 2 + 2 * v_] |= (bit_); \
        }                                                                       \
    } while (0)

    WorkCount wc{0, 0, 0, 0, 0};
    // Rays are counted per WAVE in LDS (ballot + popcount, one 64-bit LDS add by lane 0 per pass: closest | shadow << 32): no
    // register holds a count.  Camera samples are not counted at all: their number follows from the launch (host).
    uint32_t n_refills = 0;

    int state = ST_FETCH;
    uint32_t item = 0; // work items of a launch are counted in 32 bits (the host refuses more): every division below is a 32-bit one
    int px = 0, py = 0, s = 0, s_end = 0, depth = 0;
    uint32_t pixel = 0; // j * W + i of the work item's pixel: the RNG key
    bool first = true, prev_skip = false;
    PST_ST(S_ACC, mk3(0, 0, 0));  // sum over this item's samples of colour * (1/spp)
    PST_ST(S_BETA, mk3(1, 1, 1)); // path throughput
    // Across a continuation traversal the ray lives in tr.o / tr.d only.  Across a shadow traversal
    // tr.o is the shading point itself, so only the incoming direction, the hit's barycentrics and
    // the light pick need to be kept.
    d3 rd = mk3(0, 0, 1);        // incoming direction at the shading point (valid in ST_SHADOW)
    int32_t sh_tri = -1;
    real ldist = 0;              // distance to the sampled light point (valid in ST_SHADOW)
    int32_t ltri = 0;
    Rng rng;
    rng.s = 0;
    const real inv_spp = RL(1.0) / (real)P.spp; // pixelSamplesScale, Camera.cpp:83
    Trav<PAD, PRT_BOX_PK != 0> tr; // packed-FMA box test: every K3 permutation has the registers for it (coefficients in scalar registers)
    tr.init(S, mk3(0, 0, 0), rd, RL(0.0), RL(0.0));
    tr.hit.alpha = tr.hit.beta = RL(0.0); // init() leaves the barycentrics alone (they survive shadow traversals)
    tr.active = false;

#if PRT_K3_PROFILE
    unsigned long long prof_[6] = {0, 0, 0, 0, 0, 0}, prof_t_ = __builtin_readcyclecounter();
#endif
    for (;;) {
        if (COUNT) n_refills++;
        PROF_MARK(0); // traversal rounds (and loop control) since the last mark
        int started = 0; // this pass started a traversal of this kind on this lane (1 closest, 2 shadow): counted below, where the wave has reconverged
        if (!tr.active) {
            // ---------------- a traversal has just finished on this lane: consume its result
            bool end_sample = false, do_scatter = false;
            // The ray this lane traces next is written straight into tr.o / tr.d as soon as it is known (shadow ray:
            // origin = shading point, direction towards the light; continuation: same origin, scattered direction; new
            // sample: the camera ray) — no staging copies, no selects at the traversal set-up.  After a closest hit has
            // been consumed tr.o IS the shading point.
            bool have_fr = false;
            d3 fr_seen = mk3(0, 0, 0);
            if (state == ST_PRIMARY) {
                // The item's camera ray has been traced.  If it left the scene or met an emitter, every sample of the item is
                // that one radiance (Camera.cpp:127,129-132: no random number is drawn): the item's samples are added up
                // right here, in the order and with the operations ADD_RADIANCE would use one pass at a time (same bits),
                // and the lane moves on to its next item instead of sitting through one pass per sample.
                bool flat = false;
                d3 x = mk3(0, 0, 0);
                // (Measured: veach-mis -2.0 %, where many pixels look at a light or past the scene; cornell +0.7 %, bathroom2 +-0 —
                // so only the permutations with glossy materials carry it.  prt_render_samples keeps the per-sample flow.)
                if ((FEAT & (PRT_FEAT_PHONG | PRT_FEAT_CT)) && P.scramble != PRT_ITEMS_FROM_LIST) {
                    if (tr.hit.tri < 0) {
                        const ColdRenderArgs q = cold_args();
                        x = mk3(q->P.background[0], q->P.background[1], q->P.background[2]);
                        flat = true;
                    } else {
                        const DMaterial& m = MATERIAL(S.shade[tr.hit.tri].material);
                        if (m.has_emission) {
                            x = ld3(m.emission);
                            flat = true;
                        }
                    }
                }
                if (flat) {
                    const d3 term = (mk3(1, 1, 1) * x) * inv_spp; // throughput 1 at the camera vertex
                    d3 acc = PST_LD(S_ACC);
                    for (; s < s_end; ++s) acc = acc + term;
                    double* o = cold_args()->partial + (size_t)item * 3;
                    o[0] = (double)acc.x;
                    o[1] = (double)acc.y;
                    o[2] = (double)acc.z;
                    state = ST_FETCH;
                }
            }
            if (state == ST_PRIMARY) {
                // its hit serves every sample of the item (the barycentrics are only ever read for texture coordinates:
                // untextured permutations keep t and the triangle)
                park_real<NPARK>(park, PARK_T, tr.hit.t);
                park[PARK_TRI * NPARK] = (uint32_t)tr.hit.tri;
                if (FEAT & PRT_FEAT_TEX) {
                    park_real<NPARK>(park, PARK_AB_AT(FEAT), tr.hit.alpha);
                    park_real<NPARK>(park, PARK_AB_AT(FEAT) + PRT_RW, tr.hit.beta);
                }
                state = ST_NEW_SAMPLE; // set up at the bottom of this pass, consumed by the next one
            }
            if (state == ST_CACHED) state = ST_CLOSEST; // a sample set up by the previous pass from the parked hit: nothing to trace
            if (state == ST_CLOSEST) {
                if (tr.hit.tri < 0) {
                    // miss: background for the camera ray (Camera.cpp:127); with bSampleLights a bounce miss adds 0 (:187)
                    if (first || !P.sample_lights) {
                        const ColdRenderArgs q = cold_args();
                        ADD_RADIANCE(mk3(q->P.background[0], q->P.background[1], q->P.background[2]));
                    }
                    TRACE_VERTEX(-1);
                    end_sample = true;
                } else {
                    const DMaterial& m = MATERIAL(S.shade[tr.hit.tri].material);
                    TRACE_VERTEX(S.shade[tr.hit.tri].prim);
                    if (m.has_emission) {
                        // Camera.cpp:129-132; via a bounce only after SkipLightSampling materials (:191-195)
                        if (first || !P.sample_lights || prev_skip) ADD_RADIANCE(ld3(m.emission));
                        end_sample = true;
                    } else {
                        rd = tr.d;
                        sh_tri = tr.hit.tri;
                        tr.o = tr.o + tr.d * tr.hit.t; // record.position = ray(t): from here on the origin of whatever comes next
                        do_scatter = true;
                        if (P.sample_lights && S.n_lights > 0 && !m.skip_light_sampling) {
                            // next-event estimation, Camera.cpp:137-155: pick the light point now (4 draws)
                            const d3 gn = ld3(tri_at<PAD>(S, (uint32_t)sh_tri)->n);
                            const d3 fn = dot(rd, gn) < RL(0.) ? gn : -gn;
                            const LightPick lp = sample_lights<LLDS, (FEAT & PRT_FEAT_EXTRA) != 0>(S, tr.o, rng, lds_lights, P.light_lds, lds_ltris, P.ltri_lds);
                            real dist;
                            const d3 ldir = normalize_len(lp.pos - tr.o, dist);
                            if (dot(fn, ldir) > RL(0.0) && lp.front) {
                                ltri = lp.tri;
                                ldist = dist;
                                tr.d = ldir;
                                TRACE_FLAG(PRT_TRACE_NEE);
                                state = ST_SHADOW; // trace the shadow ray, then scatter
                                do_scatter = false;
                            }
                        }
                    }
                }
            } else if (state == ST_SHADOW) {
                // ---- shadow ray returned: visibility = closest hit no nearer than dist - 1e-3 (Camera.cpp:152-155)
                const real dist = ldist;
                // The shadow ray was traced over [0.001, dist - 0.001] only: the reference's test
                // `dist - |ps - pNearest| < 0.001` on the closest hit of [0.001, DBL_MAX) (|direction| = 1, so the distance
                // IS t) fails exactly when some triangle is hit inside that interval; an escaping ray counts as unoccluded (B9)
                const bool visible = tr.hit.tri < 0;
                if (visible) {
                    // the barycentrics are still the shading point's: shadow traversals leave them alone
                    const ShadeCtx c = make_ctx<FEAT, PAD>(S, rd, tr.hit.alpha, tr.hit.beta, sh_tri);
                    const DMaterial& m = MATERIAL(c.material);
                    d3 ln0;
                    real pdf;
                    int32_t lmat;
                    if (LLDS && P.ltri_lds > 0) {
                        const DLightTri* lt = lds_ltris + ltri;
                        ln0 = ld3(lt->n); pdf = lt->pdf; lmat = lt->material;
                    } else {
                        const DLightTri* lt = S.light_tris + ltri;
                        ln0 = ld3(lt->n); pdf = lt->pdf; lmat = lt->material;   // Triangle.cpp:92, BVH.cpp:91,66
                    }
                    // SetFaceNormal(Ray(origin, p - origin), normal) (Triangle.cpp:89-90); p - origin = tr.d * dist
                    const d3 ln = dot(tr.d, ln0) < RL(0.) ? ln0 : -ln0;
                    const d3 emission = ld3(MATERIAL(lmat).emission);
                    const d3 wo = world_to_local(-rd, c.f);
                    const d3 lwi = world_to_local(tr.d, c.f);
                    const d3 fr = mat_eval<FEAT>(S, m, lwi, wo, c.uv, rng);
                    if (FEAT & PRT_FEAT_TEX) {
                        have_fr = m.type == 0; // Lambertian: Eval returned albedo / pi, which Scatter needs again
                        fr_seen = fr;
                    }
                    const real cosT = lwi.z;
                    // cosThetaB = dot(WorldToLocal(lightNormal), -wi) (Camera.cpp:166-170): the shading frame is
                    // orthonormal (tangent in the triangle's plane, bitangent = t x n), so the local dot product IS the
                    // world one — three multiplies instead of a third change of basis; differs by rounding only
                    const real cosTB = -dot(ln, tr.d);
                    // Camera.cpp:172: emission*fr*cosT*cosTB/dist^2/pdf, the scalar factor folded into one division
                    const d3 direct = (emission * fr) * fast_div(cosT * cosTB, (dist * dist) * pdf);
                    ADD_RADIANCE(direct);
                    TRACE_FLAG(PRT_TRACE_VISIBLE);
                }
                state = ST_CLOSEST;
                do_scatter = true;
            }

            PROF_MARK(1); // consume: closest hit (emission, light pick) or shadow ray (light evaluation)
            if (do_scatter) {
                // ---- Russian roulette + Scatter, Camera.cpp:176-202
                end_sample = true;
                if (rng.next() < P.rr) {
                    const ShadeCtx c = make_ctx<FEAT, PAD>(S, rd, tr.hit.alpha, tr.hit.beta, sh_tri);
                    const DMaterial& m = MATERIAL(c.material);
                    d3 att, wi;
                    TRACE_FLAG(PRT_TRACE_ROULETTE);
                    if (mat_scatter<FEAT>(S, m, rd, c.f, c.uv, rng, att, wi, have_fr, fr_seen)) {
                        TRACE_FLAG(PRT_TRACE_SCATTER);
                        depth--; // RayColor(scattered, depth-1): returns 0 when depth-1 < 0
                        if (depth >= 0) {
                            const d3 beta = (PST_LD(S_BETA) * att) * P.inv_rr;
                            PST_ST(S_BETA, beta);
                            // a zero throughput (Phong bad sample) contributes exactly 0 from here on
                            if (!(beta.x == RL(0.) && beta.y == RL(0.) && beta.z == RL(0.))) {
                                tr.d = wi;
                                prev_skip = m.skip_light_sampling != 0;
                                first = false;
                                end_sample = false;
                            }
                        }
                    }
                }
            }
            PROF_MARK(2); // roulette + Scatter
            if (end_sample) {
                const d3 acc = PST_LD(S_ACC);
                s++;
                if (s < s_end) state = ST_NEW_SAMPLE;
                else {
                    double* o = cold_args()->partial + (size_t)item * 3;
                    o[0] = (double)acc.x; // the partial sums are fp64 in either mode (K5 adds them in fp64)
                    o[1] = (double)acc.y;
                    o[2] = (double)acc.z;
                    state = ST_FETCH;
                }
            }

            // ---------------- give the lane its next piece of work
            if (state == ST_FETCH) {
                const ColdRenderArgs q = cold_args(); // everything a fetch needs is read here, on demand
                const uint32_t n_items = (uint32_t)q->P.n_items;
                // One returning atomic per wave and pass on the wave's current queue (the queue index is wave-uniform,
                // so the compiler aggregates the lanes that execute it); a queue that hands out an index past the end
                // is dry for good (its indices only grow) and the lanes that drew a blank move on to the next one.
                // s_qoff[wave] = queues this wave has seen run dry.
                {
                    uint32_t off = __builtin_amdgcn_readfirstlane(s_qoff[wave]);
                    item = n_items;
                    while (off < (uint32_t)PRT_ITEM_QUEUES) {
                        const uint32_t qi = (blockIdx.x + off) % (uint32_t)PRT_ITEM_QUEUES;
                        const unsigned long long idx = atomicAdd(&ctr->queue[qi * PRT_QUEUE_STRIDE], 1ULL);
                        const unsigned long long it = ((idx >> 6) * PRT_ITEM_QUEUES + qi) * 64ULL + (idx & 63ULL);
                        if (it < n_items) {
                            item = (uint32_t)it;
                            break;
                        }
                        ++off;
                    }
                    atomicMax(&s_qoff[wave], off);
                }
                if (item >= n_items) {
                    state = ST_DONE;
                } else {
                    const uint32_t ipc = (uint32_t)q->P.items_per_chunk;
                    const uint32_t chunk = item / ipc;
                    const uint32_t oi = item - chunk * ipc;
                    const int W = q->C.width;
                    bool valid;
                    if (P.scramble == PRT_ITEMS_FROM_LIST) { // prt_render_samples: the pixels of a list
                        const int32_t pix = ctr->pixel_list[oi];
                        py = pix / W;
                        px = pix - py * W;
                        valid = true;
                        if (COUNT && ctr->trace != nullptr) ctr->trace[(size_t)item * PRT_TRACE_WORDS] = 0;
                    } else {
                        valid = tile_pixel(P.scramble, ipc, q->P.tile, q->P.tiles_x, q->P.n_tiles, q->P.rank, q->P.nranks, W, q->C.height, oi, px, py);
                    }
                    if (valid) {
                        s = q->P.chunk_begin[chunk];
                        s_end = q->P.chunk_begin[chunk + 1];
                        pixel = (uint32_t)(py * W + px);
                        PST_ST(S_ACC, mk3(0, 0, 0));
                        if (s >= s_end) {
                            double* o = q->partial + (size_t)item * 3;
                            o[0] = o[1] = o[2] = 0.0;
                        } else if (P.jitter) {
                            state = ST_NEW_SAMPLE; // a camera ray of its own per sample (below)
                        } else {
                            // Camera::GetRay (Camera.cpp:108-117), once per work item: the pixel's one camera ray.  Its direction
                            // is parked next to the hit it is about to find.
                            const d3 ps = mk3(q->C.pixel00[0], q->C.pixel00[1], q->C.pixel00[2]) + (real)px * mk3(q->C.du[0], q->C.du[1], q->C.du[2]) +
                                          (real)py * mk3(q->C.dv[0], q->C.dv[1], q->C.dv[2]);
                            tr.o = mk3(q->C.center[0], q->C.center[1], q->C.center[2]);
                            tr.d = ps - tr.o;
                            if (!PARK_DIR_GLOBAL(FEAT)) {
                                park_real<NPARK>(park, PARK_DIR, tr.d.x);
                                park_real<NPARK>(park, PARK_DIR + PRT_RW, tr.d.y);
                                park_real<NPARK>(park, PARK_DIR + 2 * PRT_RW, tr.d.z);
                            }
                            state = ST_PRIMARY;
                        }
                    }
                }
            }
            if (state == ST_NEW_SAMPLE) {
                // per-sample stream keyed (seed, j*W+i, s)
                rng.seed_keyed(P.seed_key, (uint64_t)pixel, (uint64_t)s);
                if (P.jitter) {
                    // the disabled SampleSquare() offset of Camera.cpp:110-111, drawn per sample: y first (g++ argument order);
                    // a camera ray of its own, traced like any other
                    const ColdRenderArgs q = cold_args();
                    const int W = q->C.width;
                    const int jy = (int)(pixel / (uint32_t)W), jx = (int)(pixel - (uint32_t)jy * (uint32_t)W);
                    const real fy = (real)jy + (rng.next() - RL(0.5));
                    const real fx = (real)jx + (rng.next() - RL(0.5));
                    const d3 ps = mk3(q->C.pixel00[0], q->C.pixel00[1], q->C.pixel00[2]) + fx * mk3(q->C.du[0], q->C.du[1], q->C.du[2]) +
                                  fy * mk3(q->C.dv[0], q->C.dv[1], q->C.dv[2]);
                    tr.o = mk3(q->C.center[0], q->C.center[1], q->C.center[2]);
                    tr.d = ps - tr.o;
                    state = ST_CLOSEST;
                } else {
                    // Camera::GetRay gives every sample of the pixel the same ray (Camera.cpp:53-57, no jitter): the sample starts
                    // from the parked hit of that ray.  Nothing to trace: the lane waits for the next pass, which consumes the hit
                    // (the LDS reads below have the traversal rounds in between to arrive).
                    tr.o = mk3(s_center[0], s_center[1], s_center[2]);
                    if (PARK_DIR_GLOBAL(FEAT)) { // Camera::GetRay again (same expressions as at the fetch: the same bits)
                        const ColdRenderArgs q = cold_args();
                        const d3 ps = mk3(q->C.pixel00[0], q->C.pixel00[1], q->C.pixel00[2]) + (real)px * mk3(q->C.du[0], q->C.du[1], q->C.du[2]) +
                                      (real)py * mk3(q->C.dv[0], q->C.dv[1], q->C.dv[2]);
                        tr.d = ps - tr.o;
                    } else {
                        tr.d = mk3(unpark_real<NPARK>(park, PARK_DIR), unpark_real<NPARK>(park, PARK_DIR + PRT_RW), unpark_real<NPARK>(park, PARK_DIR + 2 * PRT_RW));
                    }
                    tr.hit.t = unpark_real<NPARK>(park, PARK_T);
                    tr.hit.tri = (int32_t)park[PARK_TRI * NPARK];
                    if (FEAT & PRT_FEAT_TEX) {
                        tr.hit.alpha = unpark_real<NPARK>(park, PARK_AB_AT(FEAT));
                        tr.hit.beta = unpark_real<NPARK>(park, PARK_AB_AT(FEAT) + PRT_RW);
                    }
                    state = ST_CACHED;
                }
                PST_ST(S_BETA, mk3(1, 1, 1));
                depth = P.max_depth;
                first = true;
                prev_skip = false;
            }
            PROF_MARK(3); // end of sample, item fetch, new sample
            // ---------------- start the traversal this lane needs next
            if (state == ST_CLOSEST || state == ST_SHADOW || state == ST_PRIMARY) {
                // ONE traversal set-up for both kinds of ray (as two divergent call sites every pass executed both, one
                // after the other: -1...2 %, 8 more spilled registers).  Camera / continuation rays: Interval(0.0001, inf),
                // closest hit (Camera.cpp:125).  Shadow rays: Ray(ps, normalize(pl-ps)) (Camera.cpp:143-150); the reference
                // takes the closest hit of Interval(0.001, DBL_MAX) and calls the light visible when that hit is no nearer
                // than dist - 0.001, so only [0.001, dist - 0.001] needs tracing and ANY hit in it settles the question:
                // boxes beyond the light are culled from the start and traversal stops at the first accepted triangle.
                const bool sh_ray = state == ST_SHADOW;
                started = sh_ray ? 2 : 1;
                if (COUNT && ctr->ray_dump != nullptr) { // developer experiment: K3's own ray stream, for K1 to replay
                    const unsigned long long k = atomicAdd(&ctr->ray_dump_n, 1ULL);
                    if (k < ctr->ray_dump_cap) {
                        PrtRay r;
                        r.o[0] = (double)tr.o.x; r.o[1] = (double)tr.o.y; r.o[2] = (double)tr.o.z;
                        r.d[0] = (double)tr.d.x; r.d[1] = (double)tr.d.y; r.d[2] = (double)tr.d.z;
                        r.tmin = sh_ray ? 0.001 : 0.0001;
                        r.tmax = sh_ray ? (double)(ldist - RL(0.001)) : __builtin_huge_val();
                        static_cast<PrtRay*>(ctr->ray_dump)[k] = r;
                    }
                }
                tr.start(S, sh_ray ? RL(0.001) : RL(0.0001), sh_ray ? ldist - RL(0.001) : PRT_INF);
            }
        }
        PROF_MARK(4); // traversal set-up
        {
            const unsigned long long nc = (unsigned long long)__popcll(__ballot(started == 1)), ns = (unsigned long long)__popcll(__ballot(started == 2));
            if (lane == 0) atomicAdd(&s_rays[wave], nc | (ns << 32));
        }
        if (__ballot(state != ST_DONE) == 0ULL) break;
        // Lanes that started a sample from the parked hit have nothing to trace: with enough of them the next pass comes at
        // once (it consumes their hits and hands them real rays) instead of after traversal rounds they would sit out.
        if (wave_count(state == ST_CACHED) >= P.cached_min) continue;

        // ---------------- traversal steps until enough lanes have finished to be worth refilling
        do {
            // the interval's lower end and the any-hit rule follow from the kind of ray: not kept as traversal state
            tr.template round<COUNT>(S, stk, wc, P.leaf_batch, P.inner_min, state == ST_SHADOW ? RL(0.001) : RL(0.0001), state == ST_SHADOW);
        } while (wave_count(tr.active) > P.keep);
    }

    unsigned long long d = wave_sum((unsigned long long)wc.nodes);
    unsigned long long e = wave_sum((unsigned long long)wc.tris);
    unsigned long long f = wave_sum((unsigned long long)wc.tris_full);
    if (lane == 0) {
        const unsigned long long rays = s_rays[wave];
        atomicAdd(&ctr->rays_closest, rays & 0xffffffffULL);
        atomicAdd(&ctr->rays_shadow, rays >> 32);
        if (COUNT) {
            atomicAdd(&ctr->node_fetches, d);
            atomicAdd(&ctr->tri_full, f);
#if PRT_K3_PROFILE
            atomicAdd(&ctr->tri_tests, prof_[0]);
            atomicAdd(&ctr->inner_rounds, prof_[1]);
            atomicAdd(&ctr->leaf_rounds, prof_[2]);
            atomicAdd(&ctr->refills, prof_[3]);
            atomicAdd(&ctr->tri_full, prof_[4]);
#else
            atomicAdd(&ctr->tri_tests, e);
            atomicAdd(&ctr->inner_rounds, (unsigned long long)wc.inner_rounds);
            atomicAdd(&ctr->leaf_rounds, (unsigned long long)wc.leaf_rounds);
            atomicAdd(&ctr->refills, (unsigned long long)n_refills);
#endif
        }
    }
}

#if !PRT_F32_TU
// ------------------------------------------------------------------------------------------- K5
// out[pixel] = sum over chunks (fixed order) of the item partial sums; pixels of other ranks' tiles
// were zeroed by a memset so that the cross-rank sum is exact.
__global__ void k_finalize(DCamera C, DRenderParams P, const double* __restrict__ partial, double* __restrict__ out64,
                           float* __restrict__ out32) {
    const uint64_t oi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (oi >= P.items_per_chunk) return;
    int px, py;
    if (!owned_to_pixel(P, C, oi, px, py)) return;
    double r = 0, g = 0, b = 0;
    for (int c = 0; c < P.chunks; ++c) {
        const double* p = partial + ((uint64_t)c * P.items_per_chunk + oi) * 3;
        r += p[0];
        g += p[1];
        b += p[2];
    }
    const size_t m = ((size_t)py * C.width + px) * 3;
    if (out64) {
        out64[m] = r;
        out64[m + 1] = g;
        out64[m + 2] = b;
    }
    if (out32) {
        out32[m] = (float)r;
        out32[m + 1] = (float)g;
        out32[m + 2] = (float)b;
    }
}

__global__ void k_sample_lights(DScene S, const double* __restrict__ origins, size_t n, uint64_t seed,
                                PrtLightSample* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng rng;
    rng.seed(seed, i, 0);
    const LightPick lp = sample_lights<false, true>(S, mk3(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]), rng);
    PrtLightSample o;
    o.position[0] = lp.pos.x; o.position[1] = lp.pos.y; o.position[2] = lp.pos.z;
    o.normal[0] = lp.n.x; o.normal[1] = lp.n.y; o.normal[2] = lp.n.z;
    o.pdf = lp.pdf;
    o.prim = S.light_tris[lp.tri].prim;
    o.front = lp.front ? 1 : 0;
    out[i] = o;
}

// Test hooks for the material arithmetic (prt_material_eval / prt_material_scatter / prt_texture_value): the device
// functions K3 shades with, on caller-supplied directions; item i draws from the stream keyed (seed, i, 0).
__global__ void k_material_eval(DScene S, int material, const double* __restrict__ wi, const double* __restrict__ wo,
                                const double* __restrict__ uv, size_t n, uint64_t seed, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng rng;
    rng.seed(seed, i, 0);
    const d2 t = uv ? d2{uv[i * 2], uv[i * 2 + 1]} : d2{0., 0.};
    const d3 f = mat_eval<PRT_FEAT_ALL | PRT_FEAT_EXTRA>(S, S.materials[material], ld3(wi + i * 3), ld3(wo + i * 3), t, rng);
    out[i * 3] = f.x; out[i * 3 + 1] = f.y; out[i * 3 + 2] = f.z;
}
__global__ void k_material_scatter(DScene S, int material, const double* __restrict__ rd, d3 normal, d3 tangent,
                                   const double* __restrict__ uv, size_t n, uint64_t seed, double* __restrict__ wi_out,
                                   double* __restrict__ att_out, int32_t* __restrict__ ok_out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng rng;
    rng.seed(seed, i, 0);
    const Frame f{normal, tangent};
    const d2 t = uv ? d2{uv[i * 2], uv[i * 2 + 1]} : d2{0., 0.};
    d3 att = mk3(0, 0, 0), wi = mk3(0, 0, 0);
    const bool ok = mat_scatter<PRT_FEAT_ALL | PRT_FEAT_EXTRA>(S, S.materials[material], ld3(rd + i * 3), f, t, rng, att, wi);
    ok_out[i] = ok ? 1 : 0;
    wi_out[i * 3] = ok ? wi.x : 0.; wi_out[i * 3 + 1] = ok ? wi.y : 0.; wi_out[i * 3 + 2] = ok ? wi.z : 0.;
    att_out[i * 3] = ok ? att.x : 0.; att_out[i * 3 + 1] = ok ? att.y : 0.; att_out[i * 3 + 2] = ok ? att.z : 0.;
}
__global__ void k_texture_value(DScene S, int texture, const double* __restrict__ uv, size_t n, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const d3 c = tex_value<true>(S, texture, uv[i * 2], uv[i * 2 + 1]);
    out[i * 3] = c.x; out[i * 3 + 1] = c.y; out[i * 3 + 2] = c.z;
}

// Camera::WriteColorAttachment's per-pixel transform (Camera.cpp:279-301): NaN -> 0, LinearToSRGB
// (:214-221), clamp to [0, 0.9999], * 255 truncated to uint8.
__global__ void k_tonemap(const float* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = (double)in[i];
    if (v != v) v = 0.0;
    double sv = (v <= 0.0031308) ? 12.92 * v : 1.055 * pow(v, (1. / 2.4)) - 0.055;
    sv = sv < 0.0 ? 0.0 : (sv > 0.9999 ? 0.9999 : sv);
    out[i] = (uint8_t)(sv * 255);
}

// dst += src (fp32 framebuffers of tile shares that live on ONE device: disjoint tiles, so every element is x + 0)
__global__ void k_add_f32(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

#endif // !PRT_F32_TU

} // namespace

// ------------------------------------------------------------------------------------------- launchers
namespace PRT_NS {

// The compiled permutations: lean, textures only, Phong only, CookTorrance only, everything.  A scene gets the smallest one
// that covers its materials.
int render_permutation(int feat) {
    if (feat == 0 || feat == PRT_FEAT_TEX || feat == PRT_FEAT_PHONG || feat == PRT_FEAT_CT) return feat;
    return PRT_FEAT_ALL;
}

// Bytes of LDS a block may spend on shading tables next to its traversal stacks without costing a resident block:
// 160 KB per CU, `blocks` resident blocks (4 waves per block, 4 SIMDs per CU: blocks per CU = waves per SIMD).
int render_lds_budget(int feat, int stack_depth) {
    int blocks = render_waves(render_permutation(feat));
    if (PRT_F32_TU && stack_depth <= 32) blocks = PRT_F32_WAVES > 4 && stack_depth <= 24 ? 5 : 4; // fp32: registers allow a fourth wave per SIMD when the stacks do
    const int perm = render_permutation(feat);
    return ((160 * 1024 / blocks - (int)sizeof(uint32_t) * (stack_depth + PARK_WORDS(perm)) * PRT_BLOCK - 256) / 512) * 512; // stacks + parked camera rays
}
size_t render_table_bytes(int light_lds, int mat_lds, int ltri_lds) {
    return (size_t)light_lds * sizeof(DLightNode) + (size_t)mat_lds * sizeof(DMaterial) + (size_t)ltri_lds * sizeof(DLightTri);
}

typedef void (*RenderKernel)(RenderArgs);
// COUNT instantiations exist only with PRT_FEAT_EXTRA (statistics runs: speed is not what they are for)
template <int FEAT, bool PAD>
static RenderKernel render_kernel_feat(bool count, bool llds, bool extra) {
    constexpr int X = FEAT | PRT_FEAT_EXTRA;
    if (count) return llds ? k_render<true, X, true, PAD> : k_render<true, X, false, PAD>;
    if (extra) return llds ? k_render<false, X, true, PAD> : k_render<false, X, false, PAD>;
    return llds ? k_render<false, FEAT, true, PAD> : k_render<false, FEAT, false, PAD>;
}
template <bool PAD>
static RenderKernel render_kernel_pad(bool count, int feat, bool llds, bool extra) {
    switch (render_permutation(feat)) {
    case 0: return render_kernel_feat<0, PAD>(count, llds, extra);
    case PRT_FEAT_TEX: return render_kernel_feat<PRT_FEAT_TEX, PAD>(count, llds, extra);
    case PRT_FEAT_PHONG: return render_kernel_feat<PRT_FEAT_PHONG, PAD>(count, llds, extra);
    case PRT_FEAT_CT: return render_kernel_feat<PRT_FEAT_CT, PAD>(count, llds, extra);
    default: return render_kernel_feat<PRT_FEAT_ALL, PAD>(count, llds, extra);
    }
}
// `pad`: the scene's intersection records sit 128 bytes apart (DScene::tri_stride); `extra`: the scene has light tables or
// plain texel arrays (PRT_FEAT_EXTRA kernels)
static RenderKernel render_kernel(bool count, int feat, bool llds, bool pad, bool extra) {
    return pad ? render_kernel_pad<true>(count, feat, llds, extra) : render_kernel_pad<false>(count, feat, llds, extra);
}

static size_t stack_bytes(int stack_depth) { return PRT_DYN_STACK ? (size_t)PRT_BLOCK * stack_depth * sizeof(uint32_t) : 0; }

// Resident blocks per CU of the instantiation a launch will use (`pad`: the scene's records sit at the padded stride —
// a different function with its own register count).
int render_blocks_per_cu(bool count, int feat, size_t table_bytes, int stack_depth, bool pad, bool extra) {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_kernel(count, feat, table_bytes != 0, pad, extra), PRT_BLOCK,
                                                                table_bytes + stack_bytes(stack_depth));
    if (e != hipSuccess || nb < 1) nb = 1;
    return nb;
}

void launch_trace(const DScene& S, const PrtRay* d_rays, size_t n, PrtHit* d_hits, DCounters* d_ctr, bool count,
                  int n_cu, hipStream_t st, const uint32_t* d_perm) {
    if (n == 0) return;
    size_t want = (n + PRT_BLOCK - 1) / PRT_BLOCK;
    const size_t per_cu = std::max<size_t>(1, (160u * 1024u) / (sizeof(uint32_t) * PRT_STACK_DEPTH * PRT_BLOCK)); // LDS stacks per CU
    unsigned grid = (unsigned)std::min<size_t>(want, (size_t)n_cu * per_cu);
    const bool pad = S.tri_stride == PRT_TRI_PAD_STRIDE(real) && sizeof(DTri) != PRT_TRI_PAD_STRIDE(real);
    auto k = count ? (pad ? k_trace_closest<true, true> : k_trace_closest<true, false>)
                   : (pad ? k_trace_closest<false, true> : k_trace_closest<false, false>);
    hipLaunchKernelGGL(k, dim3(grid), dim3(PRT_BLOCK), 0, st, S, d_rays, n, d_hits, d_ctr, d_perm);
}

void launch_render(const DScene& S, const DCamera& C, const DRenderParams& P, double* d_partial, DCounters* d_ctr,
                   bool count, int feat, unsigned grid, hipStream_t st) {
    static_assert(sizeof(DMaterial) % 16 == 0, "materials are staged in 16-byte pieces");
    const size_t tables = render_table_bytes(P.light_lds, P.mat_lds, P.ltri_lds);
    const size_t dyn_lds = tables + stack_bytes(P.stack_depth);
    const bool pad = S.tri_stride == PRT_TRI_PAD_STRIDE(real) && sizeof(DTri) != PRT_TRI_PAD_STRIDE(real);
    RenderArgs A;
    A.S = S;
    A.C = C;
    A.P = P;
    A.partial = d_partial;
    A.ctr = d_ctr;
    hipLaunchKernelGGL(render_kernel(count, feat, tables != 0, pad, S.light_tab != nullptr || S.tex_compact != 0), dim3(grid), dim3(PRT_BLOCK), dyn_lds, st, A);
}

#if PRT_F32_TU
// ------------------------------------------------------------------------------------------- fp64 -> fp32 tables
// The fp32 records are derived on the device from the resident fp64 ones (which are already in BVH leaf order for
// either builder): one thread per record, each value rounded to nearest.
namespace {
__global__ void k_convert_tris(const char* __restrict__ in, uint32_t in_stride, uint32_t n, char* __restrict__ out, uint32_t out_stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* a = reinterpret_cast<const double*>(in + (size_t)i * in_stride);
    float4* o = reinterpret_cast<float4*>(out + (size_t)i * out_stride);
    static_assert(sizeof(DTriT<double>) % 32 == 0 && sizeof(DTriT<float>) * 2 == sizeof(DTriT<double>), "same fields, half the size");
    for (uint32_t k = 0; k < sizeof(DTriT<float>) / 16; ++k)
        o[k] = make_float4((float)a[4 * k], (float)a[4 * k + 1], (float)a[4 * k + 2], (float)a[4 * k + 3]);
}
__global__ void k_convert_shade(const DTriShadeT<double>* __restrict__ in, uint32_t n, DTriShadeT<float>* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DTriShadeT<double> a = in[i];
    DTriShadeT<float> o;
    for (int k = 0; k < 3; ++k) o.tangent[k] = (float)a.tangent[k];
    for (int k = 0; k < 2; ++k) {
        o.uv0[k] = (float)a.uv0[k];
        o.uv1[k] = (float)a.uv1[k];
        o.uv2[k] = (float)a.uv2[k];
    }
    o.material = a.material;
    o.prim = a.prim;
    o.pad[0] = 0.f;
    out[i] = o;
}
__global__ void k_convert_reals(const double* __restrict__ in, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}
} // namespace
void launch_convert_tris(const void* in, uint32_t in_stride, uint32_t n, void* out, uint32_t out_stride, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_convert_tris, dim3((n + 255) / 256), dim3(256), 0, st, static_cast<const char*>(in), in_stride, n, static_cast<char*>(out), out_stride);
}
void launch_convert_shade(const DTriShadeT<double>* in, uint32_t n, DTriShadeT<float>* out, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_convert_shade, dim3((n + 255) / 256), dim3(256), 0, st, in, n, out);
}
void launch_convert_reals(const double* in, size_t n, float* out, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_convert_reals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, out);
}
#endif // PRT_F32_TU

#if !PRT_F32_TU
void launch_finalize(const DCamera& C, const DRenderParams& P, const double* d_partial, double* d64, float* d32,
                     hipStream_t st) {
    unsigned grid = (unsigned)((P.items_per_chunk + 255) / 256);
    if (grid == 0) return;
    hipLaunchKernelGGL(k_finalize, dim3(grid), dim3(256), 0, st, C, P, d_partial, d64, d32);
}

void launch_sample_lights(const DScene& S, const double* d_origins, size_t n, uint64_t seed, PrtLightSample* d_out,
                          hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_sample_lights, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, d_origins, n, seed, d_out);
}

void launch_material_eval(const DScene& S, int material, const double* wi, const double* wo, const double* uv, size_t n,
                          uint64_t seed, double* out, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_material_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, material, wi, wo, uv, n, seed, out);
}
void launch_material_scatter(const DScene& S, int material, const double* rd, const double* normal, const double* tangent,
                             const double* uv, size_t n, uint64_t seed, double* wi_out, double* att_out, int32_t* ok_out,
                             hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_material_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, material, rd,
                       d3{normal[0], normal[1], normal[2]}, d3{tangent[0], tangent[1], tangent[2]}, uv, n, seed, wi_out, att_out, ok_out);
}
void launch_texture_value(const DScene& S, int texture, const double* uv, size_t n, double* out, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_texture_value, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S, texture, uv, n, out);
}

void launch_add_f32(float* dst, const float* src, size_t n, hipStream_t st) {
    if (n) hipLaunchKernelGGL(k_add_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, n);
}

void launch_tonemap(const float* d_in, size_t n, uint8_t* d_out, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_in, n, d_out);
}
#endif // !PRT_F32_TU

} // namespace PRT_NS

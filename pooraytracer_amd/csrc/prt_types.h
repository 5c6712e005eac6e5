// prt_types.h — device-resident data layouts shared by the host builder and the HIP kernels.
//
// HBM layout (all arrays are plain hipMalloc allocations, 128-byte aligned):
//   DNode   [n_nodes]   64 B  4-wide BVH node: four children's boxes on a 16-bit grid over the scene bounds, rounded
//                             OUTWARD (conservative cull only; every accept/reject of a hit is fp64) + 4 refs
//   DTri    [n_tris]    96 B  fp64 intersection record in BVH leaf order (Triangle.cpp:54-83 inputs; 128-byte stride for HBM-resident scenes)
//   DTriShade[n_tris]   96 B  fp64 shading record in the same order (tangent, texcoords, material)
//   DMaterial[n_mat]          material table (Material.h parameters)
//   DLightNode/DLightTri      the reference's area-CDF light tree (BVH.cpp:86-100), exact fp64 areas
//   texels_lin / DTexture     per texel cell the four bilinear taps (GetPixel() results, sRGB->linear applied on the host; 16 reals) + descriptors
//
// Precision: the records that carry real numbers are templates over the scalar type R.  `double` is the reference's
// arithmetic (glm::dvec3 everywhere) and what every host-side builder fills in; `float` is the layout of the fp32 fast
// mode (PRT_PRECISION_F32: half the bytes and registers, tolerance tier 2), derived from the fp64 arrays on the
// device.  A translation unit sees the un-suffixed names (DTri, DScene, ...) instantiated for PRT_REAL (default
// double); prt_kernels_f32.hip compiles the same kernels with PRT_REAL = float.
#pragma once
#include <stdint.h>

#ifndef PRT_REAL
#define PRT_REAL double
#endif
typedef PRT_REAL prt_real;

#define PRT_BVH_WIDTH 4      // children per node (64-byte nodes, collapsed from the binary tree: measured +11...32 % over the binary node)
#ifndef PRT_STACK_DEPTH
#define PRT_STACK_DEPTH 40   // LDS traversal stack entries per lane; the builders bound the stack a traversal can need to it.
                             // 40 = 40 KB per 256-thread block: four K1 blocks fill the 160 KB of a CU exactly.  The 4-wide collapse
                             // spends up to three entries per level, so on deep trees the budget decides how wide the nodes get:
                             // 8M-triangle soup 4.73M -> 3.77M nodes, 44.9 -> 36.0 visits per ray, +21 % (32 -> 40); 48 costs K1 a wave
#endif
#define PRT_STACK_SHALLOW 32 // a launch whose tree needs at most this many entries gets 32 KB of stacks per block: room for a
                             // fourth block per CU, which the fp32 render kernels have the registers for
#define PRT_BVH2_LEVELS 30   // inner-node levels of the binary tree both builders bound their trees to
#ifndef PRT_LEAF_MAX
#define PRT_LEAF_MAX 4       // triangles per BVH leaf (leaf ref stores count-1 in 3 bits)
#endif
#define PRT_BLOCK 256        // threads per workgroup (4 wave64)
#define PRT_MAX_CHUNKS 64    // sample chunks per pixel (work items per pixel)

// 64 bytes = four 16-byte loads per node visit: the x ranges of the four children, their y ranges, their z ranges,
// the four refs.  A range is lo | hi << 16 on a 65536^3 grid over the scene bounds (coordinate = grid_origin + q * grid_step;
// boxes rounded OUTWARD: a conservative cull only, every accept/reject of a hit is the fp64 triangle test).  Unused slots hold an
// inverted range (lo = 0xffff, hi = 0) on every axis and ref = 0x80000000; slots fill from the front and every node has
// at least two children, so only slots 2 and 3 can be unused (the traversal checks their refs).
struct alignas(64) DNode {
    uint32_t bx[4], by[4], bz[4];
    int32_t ref[4]; // >=0: inner node index; <0: leaf, ~ref = (first_tri << 3) | (count-1)
};
static_assert(sizeof(DNode) == 64, "DNode must be 64 bytes");

// 96 bytes: the plane (n, D) for the interval test, then IsInterior as two edge functions.  With w = n/(n.n):
//   alpha = w . ((p - v0) x e1) = (p - v0) . (e1 x w) = p . A - a0,   A = e1 x w, a0 = v0 . A
//   beta  = w . (e0 x (p - v0)) = (p - v0) . (w x e0) = p . B - b0,   B = w x e0, b0 = v0 . B
// (scalar triple product identities of Triangle.cpp:104-107; same values up to rounding, 4 loads and 8 fp64
// operations instead of 6 loads and 24).
// In HBM the records sit DScene::tri_stride bytes apart: 96 (packed) for scenes the caches hold, 128 (one record per
// 128-byte line, none straddling two) for scenes that stream from HBM — measured: packed +3 % on the cornell frame,
// padded +6 % on the 8M-triangle soup.
template <typename R>
struct alignas(4 * sizeof(R)) DTriT {
    R n[3];   // unit geometric normal        (Triangle.cpp:19)
    R D;      // dot(normal, v0)              (Triangle.cpp:50)
    R A[3], a0;
    R B[3], b0;
};
typedef DTriT<prt_real> DTri;
static_assert(sizeof(DTriT<double>) == 96 && sizeof(DTriT<float>) == 48, "DTri must be 96 / 48 bytes");

template <typename R>
struct alignas(4 * sizeof(R)) DTriShadeT {
    R tangent[3]; // Triangle.cpp:31-46
    R uv0[2], uv1[2], uv2[2];
    int32_t material;
    int32_t prim;      // index in PrtSceneDesc order
    R pad[sizeof(R) == 8 ? 2 : 1];
};
typedef DTriShadeT<prt_real> DTriShade;
static_assert(sizeof(DTriShadeT<double>) == 96 && sizeof(DTriShadeT<float>) == 48, "DTriShade must be 96 / 48 bytes");

template <typename R>
struct alignas(16) DMaterialT {
    int32_t type;
    int32_t texture;
    R kd[3], ks[3];
    R ns, pkd, pks;
    R emission[3]; // GetEmission(): DiffuseLight radiance / Debug albedo
    R eta[3], k[3];
    R alpha_x, alpha_y;
    int32_t has_emission, skip_light_sampling;
    R inv_ns1, spec_scale; // Phong: 1 / (Ns + 1) and (Ns + 2) / (Ns + 1), divided once on the host
};
typedef DMaterialT<prt_real> DMaterial;
static_assert(sizeof(DMaterialT<double>) == 192 && sizeof(DMaterialT<float>) % 16 == 0, "materials are staged into LDS in 16-byte pieces");

#define PRT_TEX_FOOTPRINT_BUDGET (256ull << 20) // fp64 bytes of footprint records a scene may hold; a scene beyond it keeps plain texel arrays (DScene::tex_compact)
struct DTexture {
    int32_t width, height, channels;
    int32_t has_data; // 0: no texels (Value() = (0,1,1))
    uint64_t offset; // index of the texture's first real in texels_lin
};

template <typename R>
struct alignas(16) DLightNodeT {
    R left_area;   // GetArea() of the left child (BVH.cpp:93-97)
    int32_t left, right; // >=0 node, <0: ~index into light tris
};
typedef DLightNodeT<prt_real> DLightNode;
static_assert(sizeof(DLightNodeT<double>) == 16 && sizeof(DLightNodeT<float>) == 16, "light nodes are 16 bytes");

// O(1) light pick (round 4).  TraverseSample (BVH.cpp:86-100) is a monotone step function of its float `p` on every
// subtree that holds no span-1 node over a node (left == right: BVH.cpp:21-23 — there `p >= area` wraps around to the first
// triangle): float subtraction and comparison are monotone, so the leaf reached can only move right as p grows.  For such a
// subtree the host finds, by bisection over float bit patterns THROUGH THE DESCENT ITSELF, the smallest p that reaches each
// leaf (thr[i]), and the kernel replaces the descent of the subtree — 11 dependent reads per pick for a 1280-triangle
// mesh — by one bucket read (value-linear buckets, ~2 per leaf: entry = first candidate leaf + the next threshold) and,
// now and then, a short walk along thr[].  The picks are the descent's bit for bit (verified on the host at every
// threshold +- 1 ulp, every bucket edge and 2^16 random p; a scene that fails keeps the tree).  The part of the tree
// above the tables (span-1 nodes, the few nodes over the meshes) is still descended with the reference's arithmetic.
// A node ref with PRT_LIGHT_TABLE_BIT set names a table: ref & ~BIT = index of its DLightTable in DScene::light_tab.
#define PRT_LIGHT_TABLE_BIT 0x40000000
#define PRT_LIGHT_TABLE_MIN 16 // leaves a subtree needs to get a table
struct DLightTable {   // 32 bytes, at uint32 offset 8 * index of light_tab
    uint32_t first;    // CDF index of the subtree's first leaf
    uint32_t n;        // its leaves
    uint32_t thr_off;  // uint32 offset in light_tab of thr[0..n] (floats; thr[0] = 0, thr[n] = +inf)
    uint32_t bkt_off;  // uint32 offset of the bucket entries {uint32 first candidate, float next threshold}
    uint32_t n_bkt;
    float inv_w;       // bucket(p) = min(n_bkt - 1, (uint32)(p * inv_w))
    uint32_t pad_[2];
};
static_assert(sizeof(DLightTable) == 32, "DLightTable is 32 bytes");

template <typename R>
struct alignas(sizeof(R) == 8 ? 128 : 16) DLightTriT {
    R v0[3], v1[3], v2[3];
    R n[3];
    R area;
    int32_t material, prim;
    R pdf;    // (1/area)*area/total_area evaluated in that order on the host (Triangle.cpp:92, BVH.cpp:91,66)
    R pad;
};
typedef DLightTriT<prt_real> DLightTri;
static_assert(sizeof(DLightTriT<double>) == 128 && sizeof(DLightTriT<float>) % 16 == 0, "DLightTri must be 128 bytes (fp64); staged in 16-byte pieces");

template <typename R>
struct DSceneT {
    const DNode* nodes;
    const DTriT<R>* tris;
    const DTriShadeT<R>* shade;
    const DMaterialT<R>* materials;
    const DTexture* textures;
    const R* texels_lin; // linearised texels (GetPixel() of Texture.cpp:50-65 evaluated on the host) as bilinear footprints: 16 reals per cell (prt_device.h, tex_value)
    const DLightNodeT<R>* light_nodes;
    const DLightTriT<R>* light_tris;
    const uint32_t* light_tab; // DLightTable headers, then their thresholds and buckets (null: no tables)
    int32_t light_root; // ref into light tree; valid iff n_lights > 0
    int32_t n_lights;
    R light_area;  // GetArea() of the top-level lights BVHNode
    uint32_t n_nodes, n_tris;
    float slab_scale;   // E of the slab test's pad (prt_device.h, slab_axis): the largest extent of the quantisation
                        // grid (box coordinates are taken relative to its origin)
    float pad_;
    float grid_origin[3]; // box coordinate = grid_origin + q * grid_step
    float grid_step[3];
    uint32_t tri_stride;  // bytes between consecutive DTri records: sizeof(DTri), or the padded stride for HBM-resident scenes
    uint32_t tex_compact; // 0: texels_lin holds bilinear footprints (16 reals per texel cell); 1: row-major texels, 3 reals each — one
                          // layout per SCENE, so that tex_value branches on a scalar (a per-texture flag cost bathroom2 1.4 %)
};
typedef DSceneT<prt_real> DScene;
// padded record stride: one fp64 record per 128-byte line, two fp32 records per line — never one straddling two lines
#define PRT_TRI_PAD_STRIDE(R) (sizeof(DTriT<R>) <= 48 ? 64u : 128u)
#define PRT_TRI_PADDED_ABOVE (256ull << 20) // triangle bytes beyond which records are padded to 128 bytes (Infinity Cache size)

// camera state after Camera::Initialize (Camera.cpp:75-106), computed on the host
template <typename R>
struct DCameraT {
    R center[3], pixel00[3], du[3], dv[3];
    int32_t width, height;
};
typedef DCameraT<prt_real> DCamera;

template <typename R>
struct DRenderParamsT {
    int32_t spp, max_depth, sample_lights, chunks;
    R rr, inv_rr;
    R background[3];
    uint64_t seed_key; // mix64(seed + golden ratio), hashed on the host (Rng::seed_keyed)
    int32_t tile, tiles_x, tiles_y, n_tiles;
    int32_t rank, nranks, owned_tiles, jitter; // jitter: per-sample SampleSquare pixel offset (Camera.cpp:110-111)
    int32_t keep, leaf_batch, inner_min, scramble;
    int32_t cached_min, pad0_;  // K3: with at least this many lanes holding a parked primary hit (ST_CACHED) the wave runs its next pass at once
    int32_t light_lds, mat_lds; // LLDS kernels: light-tree nodes / materials staged in (dynamic) LDS by K3
    int32_t ltri_lds;           // ... and ALL light triangles (n_lights) when there are at most 32 of them, else 0
    int32_t stack_depth;        // LDS traversal stack entries per lane of this launch (dynamic LDS): PRT_STACK_DEPTH, or less for a tree that needs less
    uint64_t items_per_chunk; // owned_tiles * tile * tile
    uint64_t n_items;         // items_per_chunk * chunks
    int32_t chunk_begin[PRT_MAX_CHUNKS + 1]; // chunk c covers samples [chunk_begin[c], chunk_begin[c+1])
};
typedef DRenderParamsT<prt_real> DRenderParams;

// K3 deals its work items from PRT_ITEM_QUEUES counters instead of one: a single address takes ~90 returning
// atomics per microsecond, which the short items at the end of a launch (and every launch at low spp) exceed.
// Queue q owns the 64-item blocks b with b % PRT_ITEM_QUEUES == q, in ascending order; a workgroup starts at queue
// blockIdx % PRT_ITEM_QUEUES and moves on to the next one when its queue runs dry.
#define PRT_ITEM_QUEUES 16
#define PRT_QUEUE_STRIDE 32 // in 8-byte words: one counter per 256 bytes (different memory channels)
#define PRT_ITEMS_FROM_LIST 2 // DRenderParams::scramble: pixels come from DCounters::pixel_list (prt_render_samples)

// device-side counters, zeroed before each call
struct DCounters {
    unsigned long long next_item;
    unsigned long long rays_closest, rays_shadow, node_fetches, tri_tests, samples;
    unsigned long long inner_rounds, leaf_rounds, refills; // COUNT builds: wave-level scheduling statistics
    unsigned long long tri_full; // COUNT builds: triangle tests that fetched the whole 128-byte record
    // prt_render_samples (test hook, DRenderParams::scramble == PRT_ITEMS_FROM_LIST): work item oi renders pixel pixel_list[oi]
    // (index j*W+i); the counting instantiation also writes each item's path signature to trace[item * PRT_TRACE_WORDS]
    const int32_t* pixel_list;
    int32_t* trace;
    // developer experiment (PRT_TUNE_DUMP_RAYS, counting instantiation): every traversal K3 starts is appended here as a PrtRay
    void* ray_dump;
    unsigned long long ray_dump_cap, ray_dump_n;
    unsigned long long pad_[PRT_QUEUE_STRIDE - 15];
    unsigned long long queue[PRT_ITEM_QUEUES * PRT_QUEUE_STRIDE]; // queue[q * PRT_QUEUE_STRIDE] = next 64-item-block-local index of queue q
};

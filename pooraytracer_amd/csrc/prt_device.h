// prt_device.h — device functions of the gfx950 path tracer (wave64, fp64 arithmetic).
//
// Everything here is written for CDNA4 directly: per-lane traversal of a 4-wide BVH (64-byte nodes: four children's
// boxes on a 16-bit grid) with a lane-strided LDS stack
// (conflict-free: entry e of lane l lives at word e*64+l), 96-byte fp64 triangle records (plane + two edge functions;
// 48 bytes in the fp32 translation unit), shading in the scalar type of the translation unit.
//
// Reference behaviour followed by each function is cited as file:line of Zoz4/Pooraytracer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prt_types.h"

#define PRT_DEV __device__ __forceinline__
// The scalar type of everything that is a real number in the reference (glm::dvec3 arithmetic): double, or float in the
// translation unit of the fp32 fast mode (prt_kernels_f32.hip).  RL() types a literal: a bare `2.` in an fp32
// expression would silently promote it to fp64.
typedef prt_real real;
#define PRT_F32 (sizeof(prt_real) == 4)
#define RL(x) ((real)(x))
template <typename R> struct real4_of;
template <> struct real4_of<double> { typedef double4 type; };
template <> struct real4_of<float> { typedef float4 type; };
typedef real4_of<real>::type real4;
#define PRT_NOCUR ((int32_t)0x80000000) // Trav::cur sentinel (never a valid leaf ref: n_tris < 2^28)
#ifndef PRT_BOX_PK
#define PRT_BOX_PK 1 // K3 permutations other than the lean one: the two plane parameters of an axis through one v_pk_fma_f32
#endif
#ifndef PRT_LEAF_BATCH
#define PRT_LEAF_BATCH 32 // parked lanes that trigger a leaf round
#endif
#ifndef PRT_INNER_MIN
#define PRT_INNER_MIN 12  // ... or at most this many lanes still at inner nodes
#endif

// ------------------------------------------------------------------ dvec3 subset (glm semantics)
struct d3 {
    real x, y, z;
};
struct d2 {
    real x, y;
};
PRT_DEV d3 mk3(real x, real y, real z) { return d3{x, y, z}; }
PRT_DEV d3 ld3(const real* p) { return d3{p[0], p[1], p[2]}; }
PRT_DEV d3 operator+(d3 a, d3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PRT_DEV d3 operator-(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PRT_DEV d3 operator-(d3 a) { return {-a.x, -a.y, -a.z}; }
PRT_DEV d3 operator*(d3 a, d3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PRT_DEV d3 operator*(d3 a, real s) { return {a.x * s, a.y * s, a.z * s}; }
PRT_DEV d3 operator*(real s, d3 a) { return {s * a.x, s * a.y, s * a.z}; }
PRT_DEV d3 operator/(d3 a, real s); // = a * (1/s): defined below, next to the reciprocal it uses
PRT_DEV real dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PRT_DEV d3 cross(d3 x, d3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
// fp64 square root / reciprocal for the shading code, where the kernels are VALU-bound: the hardware estimate
// (v_rsq_f64 / v_rcp_f64, ~2^-23 relative) plus one coupled Newton step and one residual correction — within an ulp
// or two of the IEEE result (the parity tolerance is 1e-9) in 7-8 instructions instead of the 14-17 of the
// correctly rounded, range-scaled library sequences.  Arguments are positive normal numbers wherever these are
// used (squared lengths, pdfs, |direction components|); zero is handled, NaN propagates.
PRT_DEV double fast_rsqrt(double x) { // 1 / sqrt(x), x > 0
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    const double h1 = fma(h, r, h), g1 = fma(g, r, g); // h1 ~ 0.5 / sqrt(x), g1 ~ sqrt(x)
    const double e = fma(-h1, g1, 0.5);
    return 2.0 * fma(h1, e, h1);
}
PRT_DEV double fast_sqrt(double x) { // sqrt(x), x >= 0
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    const double g1 = fma(g, r, g), h1 = fma(h, r, h);
    const double d = fma(-g1, g1, x);
    const double s = fma(d, h1, g1);
    return x == 0.0 ? 0.0 : s;
}
PRT_DEV double fast_rcp(double b) { // 1 / b, b != 0
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    return r;
}
PRT_DEV double fast_div(double a, double b) { // a / b, b != 0
    const double r = fast_rcp(b);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
// fp32 fast mode: the hardware instructions themselves (v_rsq_f32 / v_sqrt_f32 / v_rcp_f32, 1 ulp) — one instruction each
PRT_DEV float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
PRT_DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
PRT_DEV float fast_rcp(float b) { return __builtin_amdgcn_rcpf(b); }
PRT_DEV float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
// sqrt / division where the fp64 path wants the IEEE result (bit-exact light pick, triangle parameter t): exact in
// fp64, the hardware estimate in fp32
PRT_DEV double ieee_sqrt(double x) { return sqrt(x); }
PRT_DEV float ieee_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
PRT_DEV double ieee_div(double a, double b) { return a / b; }
PRT_DEV float ieee_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
PRT_DEV d3 operator/(d3 a, real s) { // glm divides each component; one reciprocal and three multiplies differ by rounding only
    const real r = fast_rcp(s);
    return {a.x * r, a.y * r, a.z * r};
}
PRT_DEV real length(d3 v) { return fast_sqrt(dot(v, v)); }
PRT_DEV d3 normalize(d3 v) { return v * fast_rsqrt(dot(v, v)); } // glm::normalize = v * inversesqrt(dot(v, v))
// normalize(v) and length(v) from one square root — glm::normalize / glm::length up to rounding
PRT_DEV d3 normalize_len(d3 v, real& len) {
    const real q = dot(v, v);
    const real inv = fast_rsqrt(q);
    len = q * inv;
    return v * inv;
}

#define PRT_PI RL(3.14159265358979323846)
#define PRT_INV_PI RL(0.31830988618379067154)
#define PRT_INV_2PI RL(0.15915494309189533577)
#define PRT_PI_OVER_2 RL(1.57079632679489661923)
#define PRT_PI_OVER_4 RL(0.78539816339744830961)
#define PRT_INF RL(__builtin_huge_val())

// ------------------------------------------------------------------ keyed counter RNG
// Replaces the reference's global std::rand() (RandomNumberGenerator.h:16-19) by a stream keyed on
// (seed, pixel, sample): the seed is hashed on its own (once per launch, on the host: seed_key()), the result is combined
// with (pixel, sample) and hashed again into the state of a xoroshiro64* generator; each draw yields 31 bits so
// xi = r / 2^31 has rand()'s granularity.  Hashing the seed separately keeps the streams of different seeds unrelated:
// folded linearly into one key, seed s+1 reused seed s's streams with neighbouring sample indices swapped.
PRT_DEV uint64_t mix64(uint64_t z) {
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}
PRT_DEV uint64_t seed_key(uint64_t seed) { return mix64(seed + 0x9E3779B97F4A7C15ULL); } // prt_host.h has the host twin
struct Rng {
    uint64_t s;
    // `key` = seed_key(seed); pixel and sample are below 2^32 (Stafford's mix13 avalanches every input bit)
    PRT_DEV void seed_keyed(uint64_t key, uint64_t pixel, uint64_t sample) {
        s = mix64(((uint64_t)((uint32_t)(key >> 32) ^ (uint32_t)(pixel + 1)) << 32) | ((uint32_t)key ^ (uint32_t)(sample + 1)));
        if (s == 0) s = 0x9E3779B97F4A7C15ULL; // the all-zero state is the generator's fixed point
    }
    PRT_DEV void seed(uint64_t seed, uint64_t pixel, uint64_t sample) { seed_keyed(seed_key(seed), pixel, sample); }
    // xoroshiro64* started from the hashed key, top 31 bits: one quarter-rate 32-bit multiply per number instead of the
    // two 64-bit multiplies (eight quarter-rate instructions) of a splitmix hash per number — with the kernels VALU-bound
    // the cheaper stream is worth +4 % (measured against a trivial LCG: +5 %).  Same stream in oracle/pt_oracle.cpp.
    PRT_DEV real next() {
        uint32_t s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32);
        const uint32_t r = s0 * 0x9E3779BBu;
        s1 ^= s0;
        s0 = __builtin_amdgcn_alignbit(s0, s0, 6) ^ s1 ^ (s1 << 9); // rotl(s0, 26)
        s1 = __builtin_amdgcn_alignbit(s1, s1, 19);                 // rotl(s1, 13)
        s = ((uint64_t)s1 << 32) | s0;
        if (PRT_F32) return (real)(r >> 8) * RL(5.9604644775390625e-8); // 24 bits: every value is a float below 1
        return (real)(r >> 1) * (RL(1.0) / RL(2147483648.0));
    }
};

// ------------------------------------------------------------------ BVH traversal
struct HitInfo {
    real t;      // closest accepted t (== tmax on miss)
    real alpha, beta;
    int32_t tri;   // BVH-order triangle index, -1 on miss
};

struct WorkCount {
    uint32_t nodes, tris;
    uint32_t tris_full; // COUNT builds: triangle tests that went past the plane / interval check (second 64 bytes fetched)
    uint32_t inner_rounds, leaf_rounds; // COUNT builds: wave-level executions of inner_step / leaf_step (lane 0 counts)
};

// Record i of the intersection array.  PAD: the records sit 128 bytes apart (one per line; scenes that stream from
// HBM) instead of sizeof(DTri) — a kernel template parameter, so the stride is a constant of each instantiation
// (as a run-time value it cost the register-starved textured permutation 3 %).
template <bool PAD>
PRT_DEV const DTri* tri_at(const DScene& S, uint32_t i) {
    return reinterpret_cast<const DTri*>(reinterpret_cast<const char*>(S.tris) + (size_t)i * (PAD ? PRT_TRI_PAD_STRIDE(real) : (uint32_t)sizeof(DTri)));
}

// Triangle::Hit + IsInterior (Triangle.cpp:54-83,100-113), inclusive interval: the reference's quantities through
// precomputed edge functions (see DTri).
PRT_DEV bool tri_test(const DTri* __restrict__ T, d3 o, d3 d, real tmin, real tmax, real& t_out, real& a_out,
                      real& b_out, const real4* pre = nullptr, uint32_t* n_full = nullptr) {
    const real4* q = reinterpret_cast<const real4*>(T);
    real4 q0 = pre ? *pre : q[0], q1 = q[1]; // `pre`: the record's first 32 bytes, fetched by the caller ahead of time
    d3 n = mk3(q0.x, q0.y, q0.z);
    real denom = dot(n, d);
    if (fabs(denom) < RL(1e-8)) return false;
    real t = ieee_div(q0.w - dot(n, o), denom);
    if (!(tmin <= t && t <= tmax)) return false;
    if (n_full) ++*n_full;
    const real4 q2 = q[2];
    const d3 p = o + d * t;
    const real alpha = (p.x * q1.x + p.y * q1.y + p.z * q1.z) - q1.w;
    const real beta = (p.x * q2.x + p.y * q2.y + p.z * q2.z) - q2.w;
    if (alpha != alpha || beta != beta) return false;
    if ((alpha < 0) || (beta < 0) || (alpha + beta > 1)) return false;
    t_out = t;
    a_out = alpha;
    b_out = beta;
    return true;
}

// Conservative double -> float conversions (round-to-nearest error <= 2^-24 relative, widened by 2^-22).
PRT_DEV float f32_up(real x) {
    const float f = (float)x;
    return f + fabsf(f) * 2.4e-7f;
}
PRT_DEV float f32_down(real x) {
    const float f = (float)x;
    return f - fabsf(f) * 2.4e-7f;
}

// Per-ray constants of the fp32 slab test.  A box plane at coordinate b = g0 + q*gs (q = the node's
// 16-bit grid index; gs = 1, g0 = 0 for fp32 nodes) is crossed at t(q) = fma(q, idq, c) with
// idq = gs*id, id ~ 1/d (v_rcp_f32, 1 ulp), c = r*id, r = g0 - o: the ray origin RELATIVE TO THE GRID ORIGIN, subtracted
// in the precision the origin comes in and rounded to fp32 once — so the rounding of the test scales with the size of
// the scene and the ray's distance from it, not with where the scene sits in the world (a unit scene at coordinate 1e6
// is culled as well as one at the origin).  The roundings of r, 1/d, gs*id, the product and the fma sum to
// <= 6 * 2^-24 * (|r| + E) * |id| in t (E = the grid's largest extent, or the largest |coordinate| for fp32 nodes);
// pad = 2^-20 * (|r| + E) * |id| covers them, subtracted on the entry plane and added on the exit plane.  |id| is
// clamped to 1e28 so a zero direction component never yields inf - inf.
struct SlabAxis {
    float idq, c_lo, c_hi;
    uint32_t rot; // 16 when the ray runs against this axis: the packed (lo | hi << 16) range is rotated so that its
                  // low half is always the ENTRY plane and c_lo / c_hi are the entry / exit constants
};
PRT_DEV SlabAxis slab_axis(real o, real d, float E, float g0, float gs) {
    SlabAxis a;
    const float df = (float)d;
    float id = __builtin_amdgcn_rcpf(df);
    if (!(fabsf(id) <= 1e28f)) id = copysignf(1e28f, df);
    const float r = (float)((real)g0 - o);
    const float c = r * id;
    const float pad = fmaf(E, fabsf(id), fabsf(c)) * 9.5367432e-7f; // (|r| + E) * |id| * 2^-20 with |c| = |r| |id|
    a.idq = gs * id;
    a.c_lo = c - pad; // entry plane
    a.c_hi = c + pad; // exit plane
    a.rot = id >= 0.f ? 0u : 16u;
    return a;
}

// Closest hit in [tmin, tmax] (replaces world.Hit: HittableList.h:26-39 -> BVH.cpp:51-61 -> AABB.cpp:38-64)
// as a resumable per-lane traversal: init() starts a ray, step() executes ONE node visit or ONE leaf,
// `active` drops when the traversal is complete.  Keeping the traversal resumable lets a wave leave
// the stepping loop as soon as enough of its lanes have finished (ballot/popcount), hand those lanes
// their next ray, and come back — lanes never idle for the slowest ray of the wave.
//
// The result does not depend on the tree except for exact ties: the closest accepted triangle is returned, and of
// two triangles hit at bit-identical t the later-TESTED one wins (inclusive interval, as in the reference) — which
// one is tested later depends on this tree's visit order, not on the reference's ("Exact ties").  Box tests are fp32 and strictly conservative (they can only fail to
// cull); every accept/reject of a hit is the fp64 triangle test.  `any_hit`: traversal stops at the first accepted
// triangle (K3's shadow rays, traced over [0.001, dist - 0.001]: anything in there is an occluder);
// -inf for closest-hit.  The stack lives in LDS, lane-strided (`stk` = this lane's column, stride 64).
template <bool PAD, bool PK = false>
struct Trav {
    d3 o, d;
    real tmin; // K1 only: K3 derives it (and the any-hit flag) from the kind of ray, see test_leaf
    HitInfo hit;
    SlabAxis ax, ay, az;
    float tminf, tbestf;
    int32_t cur;  // >= 0 inner node, < 0 leaf ref (the lane is parked there until the wave's next leaf round)
    int32_t sp;
    bool active;

    PRT_DEV void init(const DScene& S, d3 o_, d3 d_, real tmin_, real tmax_) {
        o = o_;
        d = d_;
        start(S, tmin_, tmax_);
    }
    // Starts the traversal of the ray already in o / d.
    PRT_DEV void start(const DScene& S, real tmin_, real tmax_) {
        tmin = tmin_;
        ax = slab_axis(o.x, d.x, S.slab_scale, S.grid_origin[0], S.grid_step[0]);
        ay = slab_axis(o.y, d.y, S.slab_scale, S.grid_origin[1], S.grid_step[1]);
        az = slab_axis(o.z, d.z, S.slab_scale, S.grid_origin[2], S.grid_step[2]);
        tminf = f32_down(tmin_);
        tbestf = f32_up(tmax_);
        hit.t = tmax_;
        hit.tri = -1;
        sp = 0;
        cur = 0;
        active = S.n_tris != 0;
    }

    // Entry / exit parameters of one child box: its three packed (lo | hi << 16) grid ranges against the ray's slabs.
    PRT_DEV void box4(uint32_t x, uint32_t y, uint32_t z, float& n, float& f) const {
        // no min/max per axis: after the rotation the low half IS the entry plane.  An unused slot (inverted range on
        // every axis) comes out with entry beyond exit for every ray near the scene (see inner_step for the others).
        x = __builtin_amdgcn_alignbit(x, x, ax.rot);
        y = __builtin_amdgcn_alignbit(y, y, ay.rot);
        z = __builtin_amdgcn_alignbit(z, z, az.rot);
        float nx, fx, ny, fy, nz, fz;
        if (PK) {
        // entry and exit plane of an axis in one packed FMA (v_pk_fma_f32: two fp32 FMAs per instruction, 12 instead
        // of 24 per node visit).  Costs three registers (the splat of each axis's 1/d); no gain in issue slots
        // (v_pk_fma_f32 is half rate), so it pays only through registers: K3 uses it, K1 (-1 %) does not.
        typedef float f2_ __attribute__((ext_vector_type(2)));
        const f2_ tx = __builtin_elementwise_fma(f2_{(float)(x & 0xffffu), (float)(x >> 16)}, f2_{ax.idq, ax.idq}, f2_{ax.c_lo, ax.c_hi});
        const f2_ ty = __builtin_elementwise_fma(f2_{(float)(y & 0xffffu), (float)(y >> 16)}, f2_{ay.idq, ay.idq}, f2_{ay.c_lo, ay.c_hi});
        const f2_ tz = __builtin_elementwise_fma(f2_{(float)(z & 0xffffu), (float)(z >> 16)}, f2_{az.idq, az.idq}, f2_{az.c_lo, az.c_hi});
        nx = tx.x, fx = tx.y, ny = ty.x, fy = ty.y, nz = tz.x, fz = tz.y;
        } else {
        nx = fmaf((float)(x & 0xffffu), ax.idq, ax.c_lo), fx = fmaf((float)(x >> 16), ax.idq, ax.c_hi);
        ny = fmaf((float)(y & 0xffffu), ay.idq, ay.c_lo), fy = fmaf((float)(y >> 16), ay.idq, ay.c_hi);
        nz = fmaf((float)(z & 0xffffu), az.idq, az.c_lo), fz = fmaf((float)(z >> 16), az.idq, az.c_hi);
        }
        n = fmaxf(fmaxf(fmaxf(nx, ny), nz), tminf);
        f = fminf(fminf(fminf(fx, fy), fz), tbestf);
    }
    // One visit of a 4-wide node: four 16-byte loads (x ranges, y ranges, z ranges, refs), four box tests, the
    // children that are hit sorted by entry distance (5 compare-exchanges on (distance bits, ref) pairs; a miss
    // or an unused slot sorts last), nearest one next, the others pushed far to near.  The builders guarantee that
    // the stack cannot overflow (bvh_build.cpp).
    template <bool COUNT>
    PRT_DEV void inner_step(const DScene& S, uint32_t* stk, WorkCount& wc) {
        const uint4* np = reinterpret_cast<const uint4*>(S.nodes + cur);
        const uint4 bx = np[0], by = np[1], bz = np[2], rf = np[3];
        if (COUNT) wc.nodes++;
        float n0, f0, n1, f1, n2, f2, n3, f3;
        box4(bx.x, by.x, bz.x, n0, f0);
        box4(bx.y, by.y, bz.y, n1, f1);
        box4(bx.z, by.z, bz.z, n2, f2);
        box4(bx.w, by.w, bz.w, n3, f3);
        // entry distances are >= tminf; as unsigned integers positive floats order like the floats themselves
        // (a non-positive tmin only costs ordering quality, never correctness)
        // Both builders fill a node's slots from the front and every node has at least two children, so only slots 2
        // and 3 can be unused (ref 0x80000000, inverted range).  The inverted range alone does not keep a ray out of
        // them: the pad grows with the ray's distance from the scene, and from ~2^19 scene sizes away it exceeds the
        // whole grid — entry <= exit then holds for the empty slot as well, and a traversal that descended into it
        // would lose its stack or never end.  Hence the two explicit checks.
        uint32_t k0 = n0 <= f0 ? __float_as_uint(n0) : 0xffffffffu;
        uint32_t k1 = n1 <= f1 ? __float_as_uint(n1) : 0xffffffffu;
        uint32_t k2 = (n2 <= f2 && rf.z != 0x80000000u) ? __float_as_uint(n2) : 0xffffffffu;
        uint32_t k3 = (n3 <= f3 && rf.w != 0x80000000u) ? __float_as_uint(n3) : 0xffffffffu;
        uint32_t r0 = rf.x, r1 = rf.y, r2 = rf.z, r3 = rf.w;
#define PRT_CE(ka, ra, kb, rb)                  \
    {                                           \
        const bool sw_ = kb < ka;               \
        const uint32_t tk_ = sw_ ? kb : ka;     \
        kb = sw_ ? ka : kb;                     \
        ka = tk_;                               \
        const uint32_t tr_ = sw_ ? rb : ra;     \
        rb = sw_ ? ra : rb;                     \
        ra = tr_;                               \
    }
        PRT_CE(k0, r0, k1, r1)
        PRT_CE(k2, r2, k3, r3)
        PRT_CE(k0, r0, k2, r2)
        PRT_CE(k1, r1, k3, r3)
        PRT_CE(k1, r1, k2, r2)
#undef PRT_CE
        if (k3 != 0xffffffffu) {
            stk[sp * 64] = r3;
            sp++;
        }
        if (k2 != 0xffffffffu) {
            stk[sp * 64] = r2;
            sp++;
        }
        if (k1 != 0xffffffffu) {
            stk[sp * 64] = r1;
            sp++;
        }
        if (k0 != 0xffffffffu) {
            cur = (int32_t)r0;
        } else if (sp == 0) {
            cur = PRT_NOCUR;
        } else {
            sp--;
            cur = (int32_t)stk[sp * 64];
        }
        if (cur == PRT_NOCUR) active = false;
    }

    // fp64 tests of one leaf's triangles (128-byte records); returns true when an early-out hit was accepted.
    template <bool COUNT>
    PRT_DEV bool test_leaf(const DScene& S, int32_t ref, WorkCount& wc, real tmin_use, bool any_hit) {
        const uint32_t enc = ~(uint32_t)ref;
        const uint32_t first = enc >> 3, cnt = (enc & 7u) + 1u;
        bool stop = false;
        // The plane part (n, D) of the NEXT triangle is requested before the current one is tested, so its
        // latency overlaps the current test instead of adding to it (the kernels wait on memory half the time).
        // Measured: bathroom2 +4.4 %, veach-mis +2.3 %, S0 +4.5 %, S4 +5.8 %, cornell unchanged; requesting the
        // next triangle's first 64 bytes instead of 32 costs cornell 7 % (registers) and gains nothing elsewhere.
        real4 q0n = *reinterpret_cast<const real4*>(tri_at<PAD>(S, first));
        for (uint32_t i = 0; i < cnt; ++i) {
            real t, al, be;
            if (COUNT) wc.tris++;
            const real4 q0c = q0n;
            if (i + 1 < cnt) q0n = *reinterpret_cast<const real4*>(tri_at<PAD>(S, first + i + 1));
            if (tri_test(tri_at<PAD>(S, first + i), o, d, tmin_use, hit.t, t, al, be, &q0c, COUNT ? &wc.tris_full : nullptr)) {
                hit.t = t;
                if (!any_hit) { // an any-hit (shadow) traversal only reports THAT something was hit: the barycentrics of the
                    hit.alpha = al; // shading point's own hit stay where they are, for the shading that follows
                    hit.beta = be;
                }
                hit.tri = (int32_t)(first + i);
                tbestf = f32_up(t);
                if (any_hit) stop = true;
            }
        }
        return stop;
    }

    // Leaf round of this lane: the triangles of the leaf it is parked at, then pop.
    template <bool COUNT>
    PRT_DEV void leaf_step(const DScene& S, uint32_t* stk, WorkCount& wc, real tmin_use, bool any_hit) {
        bool stop = false;
        if (cur < 0 && cur != PRT_NOCUR) {
            stop = test_leaf<COUNT>(S, cur, wc, tmin_use, any_hit);
            cur = PRT_NOCUR;
        }
        if (stop) {
            active = false;
        } else if (cur == PRT_NOCUR) {
            if (sp == 0) active = false;
            else {
                sp--;
                cur = (int32_t)stk[sp * 64];
            }
        }
    }

    // One scheduling round of the wave.  Node visits are cheap (4 loads) and run whenever a lane is
    // at an inner node; leaves are expensive (8 loads per triangle) and the texture addresser charges
    // a load instruction the same whether 3 or 64 lanes execute it, so lanes that reach a leaf PARK
    // there until at least PRT_LEAF_BATCH lanes of the wave are parked (or hardly any lane is still
    // descending), and then test their triangles together.
    // `tmin_use` / `any_hit`: the interval's lower end and whether the first accepted triangle ends the traversal.  K1
    // passes its per-ray tmin; K3 derives both from the kind of ray the lane is tracing, so neither is traversal state.
    template <bool COUNT>
    PRT_DEV void round(const DScene& S, uint32_t* stk, WorkCount& wc, int leaf_batch, int inner_min, real tmin_use,
                       bool any_hit) {
        if (COUNT && __ballot(active && cur >= 0) != 0ULL) wc.inner_rounds++;
        if (active && cur >= 0) inner_step<COUNT>(S, stk, wc);
        const bool parked = active && cur < 0;                                   // cannot descend any further right now
        const bool has_leaf = active && cur < 0 && cur != PRT_NOCUR;
        const int n_parked = __popcll(__ballot(parked));
        const int n_inner = __popcll(__ballot(active && cur >= 0));
        if (n_parked >= leaf_batch || n_inner <= inner_min) {
            if (COUNT && __ballot(has_leaf) != 0ULL) wc.leaf_rounds++;
            if (has_leaf) leaf_step<COUNT>(S, stk, wc, tmin_use, any_hit);
        }
    }
};

// Number of lanes of the wave for which `p` holds.
PRT_DEV int wave_count(bool p) { return __popcll(__ballot(p)); }

// ------------------------------------------------------------------ textures (Texture.cpp:22-71)
// GetPixel (Texture.cpp:50-65).  The per-texel work of the reference — colorScale * byte, then
// SRGBToLinear for >= 3 channels — is applied once on the host (same std::pow, same doubles) and the
// device reads the linearised texel as three doubles: 2 loads per tap instead of 6.
// In HBM the texture is stored as its BILINEAR FOOTPRINTS: record (x0, y0) holds the four taps Value() blends for a lookup
// that lands in that cell — (x0,y0), (x1,y0), (x0,y1), (x1,y1) with x1 = min(x0+1, W-1), y1 = min(y0+1, H-1) — as 12
// reals padded to 16: one 128-byte line per lookup (two fp32 records per line) instead of the two to four lines the four
// taps of a row-major texel array touch.  5.3x the bytes of the texel array; only the lines a frame touches matter.
#define PRT_TEX_QUAD_REALS 16
// EXTRA (PRT_FEAT_EXTRA kernels): the scene may hold plain texel arrays; kernels for scenes that do not are compiled without that
// path (its mere presence cost the bathroom2 frame 1.3 % in a same-box A/B: the textured kernel sits at its register limit).
template <bool EXTRA>
PRT_DEV d3 tex_value(const DScene& S, int ti, real u, real v) {
    const DTexture tx = S.textures[ti];
    if (!tx.has_data) return mk3(RL(0.), RL(1.), RL(1.));
    u = fmin(fmax(u, RL(0.0)), RL(1.0)); // std::clamp
    v = fmin(fmax(v, RL(0.0)), RL(1.0));
    real x = u * (tx.width - RL(1.));
    real y = (RL(1.) - v) * (tx.height - RL(1.));
    int x0 = (int)x, y0 = (int)y;
    real fx = x - x0, fy = y - y0;
    d3 c00, c10, c01, c11;
    if (!EXTRA || !S.tex_compact) { // footprint record of cell (x0, y0): the four taps in one 128-byte line (wave-uniform choice)
        const real4* q = reinterpret_cast<const real4*>(S.texels_lin + tx.offset + (size_t)(y0 * tx.width + x0) * PRT_TEX_QUAD_REALS);
        const real4 q0 = q[0], q1 = q[1], q2 = q[2];
        c00 = mk3(q0.x, q0.y, q0.z), c10 = mk3(q0.w, q1.x, q1.y);
        c01 = mk3(q1.z, q1.w, q2.x), c11 = mk3(q2.y, q2.z, q2.w);
    } else { // plain texel arrays (a scene beyond its footprint budget): Texture.cpp:35-41's four GetPixel calls
        const int x1 = min(x0 + 1, tx.width - 1), y1 = min(y0 + 1, tx.height - 1);
        const real* base = S.texels_lin + tx.offset;
        c00 = ld3(base + ((size_t)y0 * tx.width + x0) * 3), c10 = ld3(base + ((size_t)y0 * tx.width + x1) * 3);
        c01 = ld3(base + ((size_t)y1 * tx.width + x0) * 3), c11 = ld3(base + ((size_t)y1 * tx.width + x1) * 3);
    }
    d3 c0 = c00 * (1 - fx) + c10 * fx;
    d3 c1 = c01 * (1 - fx) + c11 * fx;
    return c0 * (1 - fy) + c1 * fy;
}

// ------------------------------------------------------------------ samplers (RandomNumberGenerator.h:39-73)
// sin and cos for |x| <= pi/4: the kernel polynomials of fdlibm (k_sin.c / k_cos.c, < 1 ulp), without the range
// reduction a general sincos() carries.  The concentric map only ever needs this range.
// a * b + k for a compile-time constant k in fp64: ONE v_fma_f64 with k in a scalar register pair.  Left to itself the
// compiler keeps every polynomial coefficient of the kernel in a vector register pair for the whole launch (22 registers
// of a kernel that sits at its register limit) and evaluates a Horner step as v_mov_b64 tmp, k + v_fmac_f64 tmp, a, b —
// two vector instructions; scalar moves of the literal issue beside the vector stream.
PRT_DEV double fma_ks(double a, double b, double k) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
#else
    return __builtin_fma(a, b, k);
#endif
}
PRT_DEV float fma_ks(float a, float b, float k) { return __builtin_fmaf(a, b, k); }
// a * k + c and a * k with the constant as the multiplier
PRT_DEV double fma_sk(double a, double k, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c));
    return r;
#else
    return __builtin_fma(a, k, c);
#endif
}
PRT_DEV double mul_ks(double a, double k) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(k));
    return r;
#else
    return a * k;
#endif
}

PRT_DEV void sincos_quarter(real x, real& sn, real& cs) {
    const real z = x * x;
    if (PRT_F32) { // |x| <= pi/4: Taylor to x^9 / x^8 is below fp32 rounding
        sn = x + (x * z) * (RL(-1.6666667e-01) + z * (RL(8.3333333e-03) + z * (RL(-1.9841270e-04) + z * RL(2.7557319e-06))));
        cs = RL(1.0) + z * (RL(-0.5) + z * (RL(4.1666667e-02) + z * (RL(-1.3888889e-03) + z * RL(2.4801587e-05))));
        return;
    }
    // (each step is fma(z, inner, coefficient), as the contraction of `coefficient + z * inner` was)
    real ps = fma_ks(z, RL(1.58969099521155010221e-10), -RL(2.50507602534068634195e-08));
    ps = fma_ks(z, ps, RL(2.75573137070700676789e-06));
    ps = fma_ks(z, ps, -RL(1.98412698298579493134e-04));
    ps = fma_ks(z, ps, RL(8.33333333332248946124e-03));
    sn = x + (x * z) * fma_ks(z, ps, -RL(1.66666666666666324348e-01));
    real pc = fma_ks(z, -RL(1.13596475577881948265e-11), RL(2.08757232129817482790e-09));
    pc = fma_ks(z, pc, -RL(2.75573143513906633035e-07));
    pc = fma_ks(z, pc, RL(2.48015872894767294178e-05));
    pc = fma_ks(z, pc, -RL(1.38888888888741095749e-03));
    pc = z * fma_ks(z, pc, RL(4.16666666666666019037e-02));
    cs = RL(1.0) - (RL(0.5) * z - z * pc);
}
// sin and cos of 2*pi*u for u in [0,1): the quarter turn nearest to u is subtracted exactly, the remainder
// (|r| <= 1/8, i.e. |angle| <= pi/4) goes through the kernel polynomials, the quadrant swaps / negates.
PRT_DEV void sincos_turns(real u, real& sn, real& cs) {
    const real k = floor(RL(4.0) * u + RL(0.5));     // 0..4
    const real r = u - RL(0.25) * k;             // exact
    real s, c;
    sincos_quarter(RL(2.0) * PRT_PI * r, s, c);
    const int q = (int)k & 3;
    sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
    cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}
// sin / cos of theta = 2 pi u: the library call the reference makes in fp64; in fp32 the hardware instructions, which
// take their argument in turns (v_sin_f32 / v_cos_f32)
PRT_DEV void sincos_any(double theta, double, double& sn, double& cs) { sincos(theta, &sn, &cs); }
PRT_DEV void sincos_any(float, float u, float& sn, float& cs) {
    sn = __builtin_amdgcn_sinf(u);
    cs = __builtin_amdgcn_cosf(u);
}
PRT_DEV d2 disk_concentric(d2 u) {
    d2 off = {RL(2.) * u.x - RL(1.), RL(2.) * u.y - RL(1.)};
    if (off.x == RL(0.) && off.y == RL(0.)) return {RL(0.), RL(0.)};
    // RandomNumberGenerator.h:39-56: theta = pi/4 * (y/x), or pi/2 - pi/4 * (x/y) — i.e. sin and cos swapped
    real sn, cs;
    if (fabs(off.x) > fabs(off.y)) {
        sincos_quarter(PRT_PI_OVER_4 * fast_div(off.y, off.x), sn, cs);
        return {off.x * cs, off.x * sn};
    }
    sincos_quarter(PRT_PI_OVER_4 * fast_div(off.x, off.y), cs, sn); // cos(pi/2 - phi) = sin(phi), sin(pi/2 - phi) = cos(phi)
    return {off.y * cs, off.y * sn};
}
// SampleCosineHemisphere: glm::dvec2(RandomDouble(), RandomDouble()) as compiled by g++ (right-to-left):
// u.y = first draw, u.x = second draw.
PRT_DEV d3 cosine_hemisphere(Rng& rng) {
    real first = rng.next();
    real second = rng.next();
    d2 dd = disk_concentric(d2{second, first});
    real z = fast_sqrt(fmax(RL(0.0), RL(1.) - dd.x * dd.x - dd.y * dd.y));
    return mk3(dd.x, dd.y, z);
}

// ------------------------------------------------------------------ shading frame (Material.h:76-98)
struct Frame {
    d3 n, t; // record.normal (face-forwarded), record.tangent
};
PRT_DEV d3 world_to_local(d3 w, const Frame& f) {
    d3 bit = cross(f.t, f.n);
    return mk3(dot(w, f.t), dot(w, bit), dot(w, f.n));
}
PRT_DEV d3 local_to_world(d3 l, const Frame& f) {
    d3 bit = cross(f.t, f.n);
    return normalize(l.x * f.t + l.y * bit + l.z * f.n);
}
PRT_DEV d3 reflect_z(d3 wo) { // Reflect(wo, (0,0,1)) = -wo + 2*dot(wo,n)*n
    real dn = wo.x * RL(0.) + wo.y * RL(0.) + wo.z * RL(1.);
    d3 n2 = (RL(2.) * dn) * mk3(RL(0.), RL(0.), RL(1.));
    return -wo + n2;
}
PRT_DEV d3 reflect(d3 wo, d3 n) { return -wo + RL(2.) * dot(wo, n) * n; }

// ------------------------------------------------------------------ CookTorrance (Material.h:368-521, MaterialUtils.h)
struct Cx {
    real re, im;
};
PRT_DEV Cx cx(real r, real i = RL(0.0)) { return {r, i}; }
PRT_DEV Cx operator+(Cx a, Cx b) { return {a.re + b.re, a.im + b.im}; }
PRT_DEV Cx operator-(Cx a, Cx b) { return {a.re - b.re, a.im - b.im}; }
PRT_DEV Cx operator*(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
PRT_DEV Cx operator/(Cx a, Cx z) {
    real scale = 1 / (z.re * z.re + z.im * z.im);
    return {scale * (a.re * z.re + a.im * z.im), scale * (a.im * z.re - a.re * z.im)};
}
PRT_DEV real cnorm(Cx z) { return z.re * z.re + z.im * z.im; }
PRT_DEV Cx csqrt_(Cx z) { // MaterialUtils.h:54-65
    real n = ieee_sqrt(cnorm(z)), t1 = ieee_sqrt(RL(.5) * (n + fabs(z.re))), t2 = RL(.5) * z.im / t1;
    if (n == 0) return cx(0);
    if (z.re >= 0) return {t1, t2};
    return {fabs(t2), copysign(t1, z.im)};
}
PRT_DEV real clampd(real v, real lo, real hi) { return v < lo ? lo : (v > hi ? hi : v); }
PRT_DEV real sqr(real v) { return v * v; }
// x^y for the Phong lobe (Material.h:205,223,247,260) as exp(y*log(x)): |y*log x| <= ~50 here, so the
// result is within ~1e-14 relative of pow() — far inside the 1e-9 parity tolerance — at a third of the
// instructions and registers of the fp64 library pow.  x = 0 -> 0, x < 0 -> NaN (every caller then
// takes its 'pdf <= 0 / lobe <= 0' branch exactly as with pow's negative/NaN result).
PRT_DEV float pow_pos(float x, float y) { // fp32 fast mode: v_exp_f32(y * v_log_f32(x)) (base 2); x = 0 -> exp2(-inf) = 0
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}
// Written out (fdlibm's log and a degree-13 exp on the reduced argument, both < 2 ulp) instead of calling the library
// exp and log: a third of their instructions and, above all, ~25 fewer live registers at the Phong lobe — the
// difference between 2 and 3 resident waves per SIMD for the Phong permutation.  x is a positive normal number or 0
// wherever this is used (cosines of lobe angles, uniform draws >= 2^-31), the product y*log(x) is <= ~0.
PRT_DEV double pow_pos(double x, double y) {
    // log(x): x = m * 2^k with m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f), log(1 + f) = f - (f^2/2 - s (f^2/2 + R(s^2)))
    double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < (0.70710678118654752440);
    m = lo ? (2.0) * m : m;
    k = lo ? k - 1 : k;
    const double f = m - (1.0);
    const double s = fast_div(f, (2.0) + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma_ks(w, fma_ks(w, (1.531383769920937332e-01), (2.222219843214978396e-01)), (3.999999999940941908e-01));
    const double t2 = z * fma_ks(w, fma_ks(w, fma_ks(w, (1.479819860511658591e-01), (1.818357216161805012e-01)), (2.857142874366239149e-01)), (6.666666666666735130e-01));
    const double hfsq = (0.5) * f * f;
    const double dk = (double)k;
    // (dk * ln2_hi - X is one fma(dk, ln2_hi, -X), as the contraction of the expression was)
    const double lg = fma_sk(dk, (6.93147180369123816490e-01), -((hfsq - fma(s, hfsq + (t1 + t2), mul_ks(dk, (1.90821492927058770002e-10)))) - f));
    // exp(t): t = n ln2 + r, |r| <= ln2 / 2
    const double t = fmax(y * lg, -(1000.0));
    const double n = rint(mul_ks(t, (1.44269504088896338700e+00)));
    double r = fma_sk(n, -(6.93147180369123816490e-01), t);
    r = fma_sk(n, -(1.90821492927058770002e-10), r);
    double p = (1.6059043836821613e-10);                    // 1/13!
    p = fma_ks(p, r, (2.08767569878681e-09));                   // 1/12!
    p = fma_ks(p, r, (2.505210838544172e-08));
    p = fma_ks(p, r, (2.755731922398589e-07));
    p = fma_ks(p, r, (2.7557319223985893e-06));
    p = fma_ks(p, r, (2.48015873015873e-05));
    p = fma_ks(p, r, (1.984126984126984e-04));
    p = fma_ks(p, r, (1.388888888888889e-03));
    p = fma_ks(p, r, (8.333333333333333e-03));
    p = fma_ks(p, r, (4.1666666666666664e-02));
    p = fma_ks(p, r, (1.6666666666666666e-01));
    p = fma(p, r, (0.5));
    p = fma(p, r, (1.0));
    p = fma(p, r, (1.0));
    const double e = __builtin_amdgcn_ldexp(p, (int)n);
    return x > (0.0) ? e : (0.0);
}
PRT_DEV real fr_complex(real cosTheta_i, Cx eta) { // MaterialUtils.h:100-111
    cosTheta_i = clampd(cosTheta_i, 0, 1);
    real sin2Theta_i = 1 - sqr(cosTheta_i);
    Cx sin2Theta_t = cx(sin2Theta_i) / (eta * eta);
    Cx cosTheta_t = csqrt_(cx(1) - sin2Theta_t);
    Cx r_parl = (eta * cx(cosTheta_i) - cosTheta_t) / (eta * cx(cosTheta_i) + cosTheta_t);
    Cx r_perp = (cx(cosTheta_i) - eta * cosTheta_t) / (cx(cosTheta_i) + eta * cosTheta_t);
    return (cnorm(r_parl) + cnorm(r_perp)) / 2;
}
PRT_DEV real cos2theta(d3 w) { return sqr(w.z); }
PRT_DEV real sin2theta(d3 w) { return fmax(RL(0.), 1 - cos2theta(w)); }
PRT_DEV real tan2theta(d3 w) { return sin2theta(w) / cos2theta(w); }
PRT_DEV real cosphi(d3 w) {
    real st = ieee_sqrt(sin2theta(w));
    return (st == 0) ? 1 : clampd(w.x / st, -1, 1);
}
PRT_DEV real sinphi(d3 w) {
    real st = ieee_sqrt(sin2theta(w));
    return (st == 0) ? 0 : clampd(w.y / st, -1, 1);
}
PRT_DEV real ct_D(const DMaterial& m, d3 wm) {
    real t2 = tan2theta(wm);
    if (isinf(t2)) return 0;
    real cos4 = sqr(cos2theta(wm));
    real e = t2 * (sqr(cosphi(wm) / m.alpha_x) + sqr(sinphi(wm) / m.alpha_y));
    return 1 / (PRT_PI * m.alpha_x * m.alpha_y * cos4 * sqr(1 + e));
}
PRT_DEV real ct_lambda(const DMaterial& m, d3 w) {
    real t2 = tan2theta(w);
    if (isinf(t2)) return 0;
    real alpha2 = sqr(cosphi(w) * m.alpha_x) + sqr(sinphi(w) * m.alpha_y);
    return (ieee_sqrt(1 + alpha2 * t2) - 1) / 2;
}
PRT_DEV real ct_G1(const DMaterial& m, d3 w) { return 1 / (1 + ct_lambda(m, w)); }
PRT_DEV real ct_G(const DMaterial& m, d3 wo, d3 wi) { return 1 / (1 + ct_lambda(m, wo) + ct_lambda(m, wi)); }
PRT_DEV real ct_Dv(const DMaterial& m, d3 w, d3 wm) {
    return ct_G1(m, w) / fabs(w.z) * ct_D(m, wm) * fabs(dot(w, wm));
}
PRT_DEV d3 ct_fresnel(const DMaterial& m, d3 wo, d3 wm) {
    real c = fabs(dot(wo, wm));
    return mk3(fr_complex(c, cx(m.eta[0], m.k[0])), fr_complex(c, cx(m.eta[1], m.k[1])),
               fr_complex(c, cx(m.eta[2], m.k[2])));
}
PRT_DEV d3 ct_sample_wm(const DMaterial& m, d3 w, d2 u) { // Material.h:412-435
    d3 wh = normalize(mk3(m.alpha_x * w.x, m.alpha_y * w.y, w.z));
    if (wh.z < 0) wh = -wh;
    d3 T1 = (wh.z < RL(0.99999)) ? normalize(cross(mk3(RL(0.), RL(0.), RL(1.)), wh)) : mk3(1, 0, 0);
    d3 T2 = cross(wh, T1);
    real r = ieee_sqrt(u.x), theta = 2 * PRT_PI * u.y; // SampleUniformDiskPolar
    real sn, cs;
    sincos_any(theta, u.y, sn, cs);
    d2 p = {r * cs, r * sn};
    real h = ieee_sqrt(1 - sqr(p.x));
    real lx = (1 + wh.z) / 2;
    p.y = (1 - lx) * h + lx * p.y; // Lerp
    real pz = ieee_sqrt(fmax(RL(0.), RL(1.) - (sqr(p.x) + sqr(p.y))));
    d3 nh = p.x * T1 + p.y * T2 + pz * wh;
    return normalize(mk3(m.alpha_x * nh.x, m.alpha_y * nh.y, fmax(RL(1e-6), nh.z)));
}

// ------------------------------------------------------------------ Material::Eval for NEE
// Lambertian Material.h:128-130; Phong :227-248 (draws one uniform); CookTorrance :474-496.
// FEAT selects the kernel permutation (chosen per scene by the host from the material table):
// PRT_FEAT_TEX image textures, PRT_FEAT_PHONG PhoneReflectance, PRT_FEAT_CT CookTorrance.  What a scene
// does not use is compiled out — registers, not instructions, are what the shading code costs (a
// permutation that fits one more wave per SIMD is worth more than any instruction-level tuning).
#define PRT_FEAT_TEX 1
#define PRT_FEAT_PHONG 2
#define PRT_FEAT_CT 4
#define PRT_FEAT_ALL 7
// Not a material feature: kernels with this bit also carry the two rarely needed code paths — the light-table lookup of
// sample_lights and the plain-texel-array path of tex_value.  A scene that needs neither (no emissive subtree of >= 16
// triangles, textures within the footprint budget) runs the kernels without them (same-box A/B: cornell +0.9 %, bathroom2 +1.3 %).
#define PRT_FEAT_EXTRA 8
template <int FEAT>
PRT_DEV d3 mat_kd(const DScene& S, const DMaterial& m, d2 uv) {
    if ((FEAT & PRT_FEAT_TEX) && m.texture >= 0) return tex_value<(FEAT & PRT_FEAT_EXTRA) != 0>(S, m.texture, uv.x, uv.y);
    return ld3(m.kd);
}
template <int FEAT>
PRT_DEV d3 mat_ks(const DScene& S, const DMaterial& m, d2 uv) { // Phong(mapKd,...) stores the map in Ks too (:178-181)
    if ((FEAT & PRT_FEAT_TEX) && m.texture >= 0) return tex_value<(FEAT & PRT_FEAT_EXTRA) != 0>(S, m.texture, uv.x, uv.y);
    return ld3(m.ks);
}
template <int FEAT>
PRT_DEV d3 mat_eval(const DScene& S, const DMaterial& m, d3 wi, d3 wo, d2 uv, Rng& rng) {
    if (m.type == 0) return mat_kd<FEAT>(S, m, uv) * PRT_INV_PI;
    if ((FEAT & PRT_FEAT_PHONG) && m.type == 1) {
        real u = rng.next();
        if (u < m.pkd) {
            if (wi.z <= 0) return mk3(0, 0, 0);
            return mat_kd<FEAT>(S, m, uv) * PRT_INV_PI;
        } else if (m.pkd <= u && u < m.pkd + m.pks) {
            if (wi.z <= 0) return mk3(0, 0, 0);
            d3 lr = normalize(reflect_z(wo));
            real ca = fmax(RL(0.), dot(wi, lr));
            if (ca <= RL(0.)) return mk3(0, 0, 0);
            return mat_ks<FEAT>(S, m, uv) * (m.ns + RL(2.)) * PRT_INV_2PI * pow_pos(ca, m.ns);
        }
        return mk3(0, 0, 0);
    }
    if ((FEAT & PRT_FEAT_CT) && m.type == 3) {
        if (!(wo.z * wi.z > 0)) return mk3(0, 0, 0);
        real co = fabs(wo.z), ci = fabs(wi.z);
        if (ci == 0 || co == 0) return mk3(0, 0, 0);
        d3 wm = wi + wo;
        if (sqr(wm.x) + sqr(wm.y) + sqr(wm.z) == 0) return mk3(0, 0, 0);
        wm = normalize(wm);
        d3 F = ct_fresnel(m, wo, wm);
        return ct_D(m, wm) * F * ct_G(m, wo, wi) / (4 * ci * co);
    }
    return mk3(0, 0, 0);
}

// ------------------------------------------------------------------ Material::Scatter
// Returns false when the reference's Scatter returns false.  `wi_world` is the (normalised) scattered
// direction, `att` = f * cos / pdf.  rd = incoming ray direction (unnormalised for camera rays).
template <int FEAT>
PRT_DEV bool mat_scatter(const DScene& S, const DMaterial& m, d3 rd, const Frame& f, d2 uv, Rng& rng, d3& att,
                         d3& wi_world, bool have_fr = false, d3 fr_pre = d3{RL(0.), RL(0.), RL(0.)}) {
    if (!(FEAT & PRT_FEAT_PHONG) && m.type == 1) return false; // not reachable: the host picks a permutation that
    if (!(FEAT & PRT_FEAT_CT) && m.type == 3) return false;    // covers every material type of the scene
    switch (m.type) {
    case 0: { // Lambertian, Material.h:106-151
        d3 wi = cosine_hemisphere(rng);
        while (wi.z <= RL(0.)) wi = cosine_hemisphere(rng);
        real pdf = wi.z * PRT_INV_PI;
        // textured surfaces: the light evaluation of this vertex (same pass) has already looked the albedo up
        d3 fr = ((FEAT & PRT_FEAT_TEX) && have_fr) ? fr_pre : mat_kd<FEAT>(S, m, uv) * PRT_INV_PI;
        wi_world = local_to_world(wi, f);
        att = (fr * wi.z) * fast_rcp(pdf); // fr*cos/pdf with one reciprocal (last-bit rounding only)
        return true;
    }
    case 1: { // PhoneReflectance, Material.h:183-285
        d3 wo = world_to_local(-rd, f);
        d3 wi = mk3(0, 0, 0), fr = mk3(0, 0, 0);
        real pdf = 0;
        bool spec = false, spec_ok = false;
        real u = rng.next();
        if (u < m.pkd) {
            wi = cosine_hemisphere(rng);
            while (wi.z <= RL(0.)) wi = cosine_hemisphere(rng);
            pdf = wi.z * PRT_INV_PI;
            fr = mat_kd<FEAT>(S, m, uv) * PRT_INV_PI;
        } else if (m.pkd <= u && u < m.pkd + m.pks) {
            real u1 = rng.next(), u2 = rng.next();
            // alpha = acos(u1^(1/(Ns+1))), phi = 2 pi u2 (Material.h:205-208): cos(alpha) IS the power, sin(alpha)
            // its Pythagorean complement, and sin/cos(2 pi u2) reduce exactly in u2 to a quarter-period polynomial —
            // no acos, no general-range sincos (their results differ from these by rounding only)
            real sa, ca, sp, cp;
            ca = fmin(pow_pos(u1, m.inv_ns1), RL(1.0));
            sa = ieee_sqrt(fmax(RL(0.0), RL(1.0) - ca * ca));
            sincos_turns(u2, sp, cp);
            d3 rw = mk3(sa * cp, sa * sp, ca);
            // ReflectiveSpaceToLocal, Material.h:299-311
            d3 lr = normalize(reflect_z(wo));
            d3 V = (fabs(lr.x) > RL(0.9) ? mk3(RL(0.), RL(1.), RL(0.)) : mk3(RL(1.), RL(0.), RL(0.)));
            d3 T = normalize(cross(V, lr));
            d3 B = cross(lr, T);
            wi = rw.x * T + rw.y * B + rw.z * lr;
            // SpecularPDF (Material.h:255-261) and f (:222-224) both carry cos^Ns(alpha) of the angle to the mirror
            // direction, and cos(alpha) = dot(wi, lr) IS ca = u1^(1/(Ns+1)) (T, B, lr are orthonormal), so the power is
            // u1^(Ns/(Ns+1)): positive whenever u1 is (u1 >= 2^-31 or 0 — it cannot underflow), and it cancels in
            // f cos / pdf = Ks (Ns+2)/(Ns+1) cos(theta_i).  No second pow (one log + one exp less per specular sample).
            spec_ok = wi.z > RL(0.) && u1 > RL(0.);
            spec = true;
            fr = mat_ks<FEAT>(S, m, uv) * m.spec_scale;
        }
        wi_world = local_to_world(wi, f);
        if (spec) att = spec_ok ? fr * wi.z : mk3(0, 0, 0); // below the horizon: pdf = 0, attenuation unassigned upstream, 0 here (B13)
        else if (pdf > RL(0.) && wi.z > 0) att = fr * wi.z / pdf;
        else att = mk3(0, 0, 0); // reference leaves it unassigned (Material.h:280-282); defined 0 (B13)
        return true;
    }
    case 2: { // PerfectMirror, Material.h:334-363
        d3 wo = world_to_local(-rd, f);
        d3 wi = reflect_z(wo);
        real c = wi.z;
        d3 fr = mk3(RL(1.0) / c, RL(1.0) / c, RL(1.0) / c);
        wi_world = local_to_world(wi, f);
        att = fr * c / RL(1.0);
        return true;
    }
    case 3: { // CookTorrance, Material.h:437-516
        d3 wo = normalize(world_to_local(-rd, f));
        if (wo.z == 0) return false;
        real first = rng.next();
        real second = rng.next();
        d3 wm = ct_sample_wm(m, wo, d2{second, first});
        d3 wi = reflect(wo, wm);
        if (!(wo.z * wi.z > 0)) return false;
        real pdf = ct_Dv(m, wo, wm) / (RL(4.) * fabs(dot(wo, wm)));
        real co = fabs(wo.z), ci = fabs(wi.z);
        if (ci == 0 || co == 0) return false;
        d3 F = ct_fresnel(m, wo, wm);
        d3 fr = ct_D(m, wm) * F * ct_G(m, wo, wi) / (RL(4.) * ci * co);
        att = fr * wi.z / pdf;
        wi_world = local_to_world(wi, f);
        return true;
    }
    default: return false; // DiffuseLight / Debug / Empty: Material::Scatter base (Material.h:57-59)
    }
}

// ------------------------------------------------------------------ lights.Sample
// HittableList::Sample (HittableList.h:44-59, one discarded draw) -> BVHNode::Sample (BVH.cpp:62-67,
// p = ieee_sqrt(xi)*A truncated to float) -> TraverseSample (BVH.cpp:86-100) -> Triangle::Sample
// (Triangle.cpp:84-93).  pdf = (1/area)*area/totalArea evaluated in that order.
struct LightPick {
    d3 pos, n;
    real pdf;
    int32_t tri; // index into light_tris
    bool front;
};
// `lds_nodes` / `n_lds`: the first n_lds nodes (breadth-first numbering = the top levels of the tree) staged in LDS
// by the kernel, or null / 0.  The descent is a chain of dependent 16-byte reads, one per tree level (13 for
// veach-mis's 6400 light triangles: 11 % of its frame from L1/L2, and the ~30 KB of LDS the top 1800 nodes took cost the
// Phong permutation its third wave) — since round 4 only the few nodes above the per-mesh TABLES are descended.
// LTAB: the kernel carries the light-table lookup (PRT_FEAT_EXTRA kernels; scenes without tables run kernels without it: its
// mere presence cost the cornell frame 0.9 % in a same-box A/B).
template <bool LLDS, bool LTAB>
PRT_DEV LightPick sample_lights(const DScene& S, d3 origin, Rng& rng, const DLightNode* lds_nodes = nullptr, int32_t n_lds = 0,
                                const DLightTri* lds_tris = nullptr, int32_t n_tris_lds = 0) {
    (void)rng.next();
    real p = ieee_sqrt(rng.next()) * S.light_area; // IEEE: p is truncated to float and compared against the CDF — the pick stays bit-exact
    float pf = (float)p;
    int32_t node = S.light_root;
    // plain node refs are below PRT_LIGHT_TABLE_BIT, table refs at or above it, leaf refs negative: ONE unsigned compare per
    // level, as before the tables existed (a scene without tables pays one more compare per pick, after the loop)
    while (LTAB ? (uint32_t)node < (uint32_t)PRT_LIGHT_TABLE_BIT : node >= 0) {
        DLightNode ln;
        if (LLDS && node < n_lds) ln = lds_nodes[node];
        else ln = S.light_nodes[node];
        if ((real)pf < ln.left_area) node = ln.left;
        else {
            pf = (float)((real)pf - ln.left_area);
            node = ln.right;
        }
    }
    if (LTAB && node >= 0) {
        // A subtree on which the descent is a monotone step function of p (prt_types.h, DLightTable): one bucket read gives
        // the first leaf p can reach and the threshold of the next one; thresholds were found by bisection through this
        // very descent, so the pick is the descent's bit for bit.
        const uint4* h = reinterpret_cast<const uint4*>(S.light_tab + 8u * ((uint32_t)node & ~(uint32_t)PRT_LIGHT_TABLE_BIT));
        const uint4 h0 = h[0];                                   // first, n, thr_off, bkt_off
        const uint2 h1 = *reinterpret_cast<const uint2*>(h + 1); // n_bkt, inv_w
        const uint32_t k = (uint32_t)fminf(pf * __uint_as_float(h1.y), (float)(h1.x - 1u));
        const uint2 e = *reinterpret_cast<const uint2*>(S.light_tab + h0.w + 2u * k);
        uint32_t i = e.x;
        float next = __uint_as_float(e.y);
        while (next <= pf) {
            ++i;
            next = __uint_as_float(S.light_tab[h0.z + i + 1u]);
        }
        node = ~(int32_t)(h0.x + i);
    }
    LightPick lp;
    lp.tri = ~node;
    real x = fast_sqrt(rng.next());
    real y = rng.next();
    d3 v0, v1, v2, n;
    if (LLDS && n_tris_lds > 0) { // uniform: a scene's light triangles are staged all or not at all
        const DLightTri* lt = lds_tris + lp.tri;
        v0 = ld3(lt->v0); v1 = ld3(lt->v1); v2 = ld3(lt->v2); n = ld3(lt->n);
        lp.pdf = lt->pdf;
    } else {
        const DLightTri* lt = S.light_tris + lp.tri;
        v0 = ld3(lt->v0); v1 = ld3(lt->v1); v2 = ld3(lt->v2); n = ld3(lt->n);
        lp.pdf = lt->pdf; // (1/area)*area/total_area, evaluated in that order on the host
    }
    lp.pos = v0 * (RL(1.0) - x) + v1 * (x * (RL(1.0) - y)) + v2 * (x * y);
    d3 dir = lp.pos - origin;
    lp.front = dot(dir, n) < RL(0.);
    lp.n = lp.front ? n : -n;
    return lp;
}

// ------------------------------------------------------------------ tile <-> pixel mapping (multi-GPU sharding)
// Owned-pixel index -> pixel.  Tiles are dealt round-robin over ranks; inside a tile pixels are
// visited in 8x8 blocks so the 64 lanes of a wave start on one compact block.
PRT_DEV bool tile_pixel(int scramble, uint32_t items_per_chunk, int tile, int tiles_x, int n_tiles, int rank, int nranks, int width,
                        int height, uint32_t oi, int& px, int& py) {
    if (scramble) oi = (uint32_t)(((uint64_t)oi * 2654435761ULL) % items_per_chunk); // experiment (odd multiplier; bijective for power-of-two counts)
    const uint32_t tt = (uint32_t)(tile * tile);
    const uint32_t ot = oi / tt, w = oi - ot * tt;
    uint32_t k = (uint32_t)rank + ot * (uint32_t)nranks;
    if (k >= (uint32_t)n_tiles) return false;
    // tile slot k -> tile (tx, ty): every tile row is rotated by 3*ty so that the tiles of one rank
    // (k % nranks) form diagonals instead of fixed columns — better load balance across GPUs
    const uint32_t ty = k / (uint32_t)tiles_x, kx = k - ty * (uint32_t)tiles_x;
    const uint32_t tx = (kx + 3u * ty) % (uint32_t)tiles_x;
    uint32_t bpr = (uint32_t)tile / 8u;
    uint32_t blk = w / 64u, l = w % 64u;
    const uint32_t by = blk / bpr, bx = blk - by * bpr;
    px = (int)(tx * tile + bx * 8u + (l % 8u));
    py = (int)(ty * tile + by * 8u + (l / 8u));
    return px < width && py < height;
}
PRT_DEV bool owned_to_pixel(const DRenderParams& P, const DCamera& C, uint32_t oi, int& px, int& py) {
    return tile_pixel(P.scramble, (uint32_t)P.items_per_chunk, P.tile, P.tiles_x, P.n_tiles, P.rank, P.nranks, C.width, C.height, oi, px, py);
}

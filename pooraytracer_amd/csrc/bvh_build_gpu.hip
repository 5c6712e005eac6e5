// bvh_build_gpu.hip — BVH construction on the MI355X (the reference builds its
// tree on the CPU, Source/BVH.cpp:7-48, and the default builder here is bvh_build.cpp on the host).
//
// Same tree family as the host builder (BVH2, <= PRT_LEAF_MAX triangles per leaf, SAH with the same
// costs, depth bounded to the traversal stack, 32-byte quantised nodes), built for a machine with
// 256 CUs instead of one core:
//   1. k_centroid_bounds      scene bounds of the box centres (wave reduction + ordered-int atomics)
//   2. k_morton               63-bit Morton key of every box centre
//   3. rocprim radix sort     (key, triangle index) pairs — the only library primitive used
//   4. k_tree_leaves/level    an implicit segment tree of boxes over the sorted order: the box of ANY
//                             contiguous range of the Morton order costs O(log range) 24-byte loads
//   5. k_split_level          level-synchronous top-down build, one lane per open node: a node is a
//                             range [s,e) of the sorted order, candidate splits are 15 equal-count
//                             positions plus the highest-differing-Morton-bit position (the spatial
//                             median plane), priced with the SAH from two segment-tree range boxes.
//                             Nothing is ever re-partitioned, so a level is one launch; children are
//                             appended to the next level's queue with wave-aggregated atomics.
//   6. k_quantise             fp32 child boxes -> 16-bit grid indices, rounded outward
//      (PRT_BVH_WIDTH 4: k_heights + k_collapse_level instead — the binary tree is collapsed top-down, one launch
//       per wide level, with the same greedy largest-box rule and stack budget as bvh_build.cpp)
// Results (hits, images) do not depend on which builder made the tree — only exact ties could, and the
// child order is deterministic here as well; node *numbering* follows atomic arrival order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <limits>

#include <rocprim/device/device_radix_sort.hpp>

#include "prt_host.h"

namespace prt {
namespace {

constexpr int kMaxLevels = PRT_BVH2_LEVELS;     // inner-node levels of the binary tree, as in bvh_build.cpp
constexpr float kCostTri = 1.5f, kCostNode = 1.0f;
constexpr int kCand = 15;                       // equal-count candidate positions per node
constexpr uint32_t kNoParent = 0xffffffffu;

struct FBox {
    float lo[3], hi[3];
};
struct FNodeD { // builder-side node: both children's fp32 boxes + refs
    FBox c[2];
    int32_t ref[2];
};
struct Item { // open node of the current level
    uint32_t s, e;
    uint32_t parent; // FNodeD index whose ref[side] / c[side] describe this node (kNoParent for the root)
    uint32_t side;
};
struct BuildState {
    uint32_t n_nodes;
    uint32_t depth;
    uint32_t counts[kMaxLevels + 3]; // open nodes per level
    // centroid bounds as order-preserving uints
    uint32_t cmin[3], cmax[3];
};

__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ void box_reset(FBox& b) {
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = INFINITY;
        b.hi[a] = -INFINITY;
    }
}
__device__ __forceinline__ void box_grow(FBox& b, const FBox& o) {
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(b.lo[a], o.lo[a]);
        b.hi[a] = fmaxf(b.hi[a], o.hi[a]);
    }
}
__device__ __forceinline__ float half_area(const FBox& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

__global__ void k_init_state(BuildState* st) {
    st->n_nodes = 0;
    st->depth = 0;
    for (int i = 0; i < kMaxLevels + 3; ++i) st->counts[i] = 0;
    for (int a = 0; a < 3; ++a) {
        st->cmin[a] = 0xffffffffu;
        st->cmax[a] = 0u;
    }
}

__global__ void k_centroid_bounds(const FBox* __restrict__ boxes, uint32_t n, BuildState* st) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const FBox b = boxes[i];
        for (int a = 0; a < 3; ++a) {
            const float c = 0.5f * (b.lo[a] + b.hi[a]);
            lo[a] = fminf(lo[a], c);
            hi[a] = fmaxf(hi[a], c);
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; ++a) {
            atomicMin(&st->cmin[a], f2ord(lo[a]));
            atomicMax(&st->cmax[a], f2ord(hi[a]));
        }
}

// Extended Morton key (Vinkler, Bittner, Havran 2017): the box CENTRE interleaved x,y,z from the top bit,
// with one bit of the box SIZE (diagonal / scene diagonal) spliced in after every second xyz triple, so that
// wall-sized triangles separate from the small ones around them near the top of the order instead of
// inflating every range they fall into.  17 xyz levels (51 bits) + 8 size bits = 59 bits.
#ifndef PRT_EMC_SIZE_BITS
#define PRT_EMC_SIZE_BITS 8
#endif
#ifndef PRT_EMC_EVERY
#define PRT_EMC_EVERY 2  // a size bit before every 2nd xyz triple ...
#endif
#ifndef PRT_EMC_PHASE
#define PRT_EMC_PHASE 0  // ... starting with the very first one
#endif
#ifndef PRT_EMC_LOG
#define PRT_EMC_LOG 0
#endif
constexpr int kKeyBits = 17 * 3 + PRT_EMC_SIZE_BITS;

__global__ void k_morton(const FBox* __restrict__ boxes, uint32_t n, const BuildState* st, uint64_t* __restrict__ keys,
                         uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FBox b = boxes[i];
    uint32_t q[3];
    float diag2 = 0.f, sdiag2 = 0.f;
    for (int a = 0; a < 3; ++a) {
        const float lo = ord2f(st->cmin[a]), hi = ord2f(st->cmax[a]);
        const float ext = hi - lo;
        const float c = 0.5f * (b.lo[a] + b.hi[a]);
        float t = ext > 0.f ? (c - lo) / ext : 0.f;
        t = fminf(fmaxf(t, 0.f), 1.f);
        q[a] = (uint32_t)fminf(t * 131072.f, 131071.f); // 17 bits
        const float d = b.hi[a] - b.lo[a];
        diag2 += d * d;
        sdiag2 += ext * ext;
    }
    const float rel = sdiag2 > 0.f ? sqrtf(diag2 / sdiag2) : 0.f;
#if PRT_EMC_LOG
    // size class = number of halvings below the scene diagonal, largest first: 2^-k <= rel < 2^-(k-1) -> max-k
    const float cls = fminf(fmaxf(-log2f(fmaxf(rel, 1e-30f)), 0.f), (float)((1u << PRT_EMC_SIZE_BITS) - 1u));
    const uint32_t sz = ((1u << PRT_EMC_SIZE_BITS) - 1u) - (uint32_t)cls;
#else
    const uint32_t sz = (uint32_t)fminf(rel * (float)(1u << PRT_EMC_SIZE_BITS), (float)((1u << PRT_EMC_SIZE_BITS) - 1u));
#endif
    uint64_t key = 0;
    int sbit = PRT_EMC_SIZE_BITS - 1;
    for (int l = 16; l >= 0; --l) {
        if (((16 - l) % PRT_EMC_EVERY) == PRT_EMC_PHASE && sbit >= 0) {
            key = (key << 1) | (uint64_t)((sz >> sbit) & 1u);
            --sbit;
        }
        key = (key << 3) | (uint64_t)((((q[0] >> l) & 1u) << 2) | (((q[1] >> l) & 1u) << 1) | ((q[2] >> l) & 1u));
    }
    key <<= (sbit + 1); // unused size bits: keep the key width fixed
    keys[i] = key;
    vals[i] = i;
}

// implicit segment tree: tree[M + i] = box of sorted triangle i (empty past n), tree[k] = tree[2k] U tree[2k+1]
__global__ void k_tree_leaves(const FBox* __restrict__ boxes, const uint32_t* __restrict__ order, uint32_t n, uint32_t M,
                              FBox* __restrict__ tree) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    FBox b;
    if (i < n) b = boxes[order[i]];
    else box_reset(b);
    tree[M + i] = b;
}
__global__ void k_tree_level(FBox* __restrict__ tree, uint32_t first, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t k = first + i;
    FBox b = tree[2 * k];
    box_grow(b, tree[2 * k + 1]);
    tree[k] = b;
}

__device__ inline FBox range_box(const FBox* __restrict__ tree, uint32_t M, uint32_t s, uint32_t e) {
    FBox b;
    box_reset(b);
    uint32_t l = s + M, r = e + M;
    while (l < r) {
        if (l & 1) box_grow(b, tree[l++]);
        if (r & 1) box_grow(b, tree[--r]);
        l >>= 1;
        r >>= 1;
    }
    return b;
}

__device__ __forceinline__ int32_t leaf_ref(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

// One level of the top-down build.  `level` = depth of the nodes in `in`; levels_left inner levels may follow.
__global__ void k_split_level(const Item* __restrict__ in, Item* __restrict__ out, BuildState* st, int level,
                              FNodeD* __restrict__ nodes, const FBox* __restrict__ tree, uint32_t M,
                              const uint64_t* __restrict__ keys) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_in = st->counts[level];
    const bool active = t < n_in;
    bool inner = false;
    Item it{0, 0, kNoParent, 0};
    uint32_t mid = 0;
    FBox bl, br;
    box_reset(bl);
    box_reset(br);
    if (active) {
        it = in[t];
        const uint32_t count = it.e - it.s;
        const int levels_left = kMaxLevels - level;
        const bool force_inner = it.parent == kNoParent;
        if (force_inner || (count > 1 && levels_left > 0)) {
            const FBox box = force_inner ? range_box(tree, M, it.s, it.e) : nodes[it.parent].c[it.side];
            const float parent_area = fmaxf(half_area(box), 1e-30f);
            float best_cost = INFINITY;
            auto consider = [&](uint32_t m) {
                const FBox l = range_box(tree, M, it.s, m), r = range_box(tree, M, m, it.e);
                const float cost = kCostNode + kCostTri * (half_area(l) * (float)(m - it.s) + half_area(r) * (float)(it.e - m)) / parent_area;
                if (cost < best_cost) {
                    best_cost = cost;
                    mid = m;
                    bl = l;
                    br = r;
                }
            };
            if (count <= (uint32_t)kCand + 1) {
                for (uint32_t m = it.s + 1; m < it.e; ++m) consider(m);
            } else {
                for (int k = 1; k <= kCand; ++k) consider(it.s + (uint32_t)(((uint64_t)count * (uint64_t)k) / (uint64_t)(kCand + 1)));
                // spatial-median plane: first position whose key differs from keys[s] in the highest bit
                // in which keys[s] and keys[e-1] differ (the split of a Morton radix tree)
                const uint64_t kf = keys[it.s], kl = keys[it.e - 1];
                if (kf != kl) {
                    const int prefix = __clzll((long long)(kf ^ kl));
                    uint32_t lo = it.s, hi = it.e - 1; // keys[lo] shares > prefix bits with kf, keys[hi] does not
                    while (hi - lo > 1) {
                        const uint32_t m = lo + (hi - lo) / 2;
                        if (__clzll((long long)(kf ^ keys[m])) > prefix) lo = m;
                        else hi = m;
                    }
                    consider(hi);
                }
            }
            const bool make_leaf = !force_inner && count <= PRT_LEAF_MAX && !(best_cost < kCostTri * (float)count);
            if (!make_leaf) {
                inner = true;
                const uint64_t child_cap = (uint64_t)PRT_LEAF_MAX << (levels_left - 1);
                const uint32_t big = max(mid - it.s, it.e - mid);
                if (mid <= it.s || mid >= it.e || (uint64_t)big > child_cap) { // depth bound: balanced split
                    mid = it.s + count / 2;
                    bl = range_box(tree, M, it.s, mid);
                    br = range_box(tree, M, mid, it.e);
                }
            }
        }
        if (!inner) nodes[it.parent].ref[it.side] = leaf_ref(it.s, it.e - it.s);
        atomicMax(&st->depth, (uint32_t)level);
    }
    // wave-aggregated allocation of one node and two queue entries per inner lane
    const unsigned long long mask = __ballot(inner);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    const uint32_t total = (uint32_t)__popcll(mask);
    uint32_t base_node = 0, base_item = 0;
    if (lane == leader) {
        base_node = atomicAdd(&st->n_nodes, total);
        base_item = atomicAdd(&st->counts[level + 1], 2 * total);
    }
    base_node = __shfl(base_node, leader);
    base_item = __shfl(base_item, leader);
    if (inner) {
        const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        const uint32_t me = base_node + rank;
        FNodeD nd;
        nd.c[0] = bl;
        nd.c[1] = br;
        nd.ref[0] = nd.ref[1] = 0;
        nodes[me] = nd;
        if (it.parent != kNoParent) nodes[it.parent].ref[it.side] = (int32_t)me;
        out[base_item + 2 * rank] = Item{it.s, mid, me, 0};
        out[base_item + 2 * rank + 1] = Item{mid, it.e, me, 1};
    }
}

__global__ void k_root_box(const FBox* __restrict__ tree, FBox* out) { *out = tree[1]; }

struct Grid {
    float origin[3], step[3];
};

__device__ inline uint16_t quant_lo(const Grid& g, float v, int a) {
    const double g0 = (double)g.origin[a], gs = (double)g.step[a];
    double q = floor(((double)v - g0) / gs);
    q = fmin(65535.0, fmax(0.0, q));
    while (q > 0 && g0 + q * gs > (double)v) q -= 1;
    return (uint16_t)q;
}
__device__ inline uint16_t quant_hi(const Grid& g, float v, int a) {
    const double g0 = (double)g.origin[a], gs = (double)g.step[a];
    double q = ceil(((double)v - g0) / gs);
    q = fmin(65535.0, fmax(0.0, q));
    while (q < 65535 && g0 + q * gs < (double)v) q += 1;
    return (uint16_t)q;
}

// Heights of the binary subtrees, one launch per level from the deepest up (nodes of one level are contiguous:
// the split launches allocate them level by level).
__global__ void k_heights(const FNodeD* __restrict__ fn, uint32_t first, uint32_t count, uint8_t* __restrict__ h2) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const FNodeD& f = fn[first + t];
    const int a = f.ref[0] >= 0 ? h2[f.ref[0]] : 0, b = f.ref[1] >= 0 ? h2[f.ref[1]] : 0;
    h2[first + t] = (uint8_t)(1 + max(a, b));
}

struct Open { // binary node `bin` becomes wide node `slot`; its subtree may use `budget` stack entries
    int32_t bin;
    uint32_t slot;
    int32_t budget;
};
struct WideState {
    uint32_t n_wide;
    uint32_t depth;
    uint32_t counts[kMaxLevels + 2]; // open nodes per wide level
};

// One level of the collapse (see bvh_build.cpp for the rule): every open binary node absorbs the child with the
// largest box while the stack budget allows, up to four children; its inner children get consecutive wide slots.
__global__ void k_collapse_level(const Open* __restrict__ in, Open* __restrict__ out, WideState* ws, int level,
                                 const FNodeD* __restrict__ fn, const uint8_t* __restrict__ h2, Grid g, DNode* __restrict__ wide) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ws->counts[level]) return;
    const Open o = in[t];
    int32_t ref[4];
    FBox box[4];
    int nk = 2;
    ref[0] = fn[o.bin].ref[0]; box[0] = fn[o.bin].c[0];
    ref[1] = fn[o.bin].ref[1]; box[1] = fn[o.bin].c[1];
    auto height = [&](int32_t r) { return r >= 0 ? (int)h2[r] : 0; };
    while (nk < 4) {
        int best = -1;
        float best_area = -1.f;
        const int left = o.budget - nk; // absorbing makes nk+1 children; each keeps budget - nk entries
        for (int i = 0; i < nk; ++i) {
            if (ref[i] < 0) continue;
            bool fits = height(fn[ref[i]].ref[0]) <= left && height(fn[ref[i]].ref[1]) <= left;
            for (int j = 0; j < nk && fits; ++j)
                if (j != i && height(ref[j]) > left) fits = false;
            const float a = half_area(box[i]);
            if (fits && a > best_area) {
                best_area = a;
                best = i;
            }
        }
        if (best < 0) break;
        const FNodeD f = fn[ref[best]];
        ref[best] = f.ref[0]; box[best] = f.c[0];
        ref[nk] = f.ref[1]; box[nk] = f.c[1];
        ++nk;
    }
    int n_inner = 0;
    for (int i = 0; i < nk; ++i) n_inner += ref[i] >= 0;
    uint32_t base = 0, qbase = 0;
    if (n_inner) {
        base = atomicAdd(&ws->n_wide, (uint32_t)n_inner);
        qbase = atomicAdd(&ws->counts[level + 1], (uint32_t)n_inner);
        atomicMax(&ws->depth, (uint32_t)level + 1u);
    }
    DNode d;
    for (int i = 0; i < 4; ++i) {
        d.bx[i] = d.by[i] = d.bz[i] = 0x0000ffffu;
        d.ref[i] = (int32_t)0x80000000;
    }
    int k = 0;
    for (int i = 0; i < nk; ++i) {
        d.bx[i] = (uint32_t)quant_lo(g, box[i].lo[0], 0) | ((uint32_t)quant_hi(g, box[i].hi[0], 0) << 16);
        d.by[i] = (uint32_t)quant_lo(g, box[i].lo[1], 1) | ((uint32_t)quant_hi(g, box[i].hi[1], 1) << 16);
        d.bz[i] = (uint32_t)quant_lo(g, box[i].lo[2], 2) | ((uint32_t)quant_hi(g, box[i].hi[2], 2) << 16);
        if (ref[i] < 0) d.ref[i] = ref[i];
        else {
            d.ref[i] = (int32_t)(base + k);
            out[qbase + k] = Open{ref[i], base + (uint32_t)k, o.budget - (nk - 1)};
            ++k;
        }
    }
    wide[o.slot] = d;
}

struct Scratch { // frees every temporary on every exit path
    std::vector<void*> p;
    ~Scratch() {
        for (void* q : p) (void)hipFree(q);
    }
    template <typename T>
    hipError_t alloc(T** out, size_t count) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, std::max<size_t>(count * sizeof(T), 256));
        if (e == hipSuccess) p.push_back(q);
        *out = static_cast<T*>(q);
        return e;
    }
};

} // namespace

namespace {
// out[i] = in[order[i]] for the 128-byte intersection and 96-byte shading records, 16 bytes per lane
__global__ void k_gather_tris(const uint4* __restrict__ tri_in, const uint4* __restrict__ shade_in,
                              const uint32_t* __restrict__ order, uint32_t n, uint4* __restrict__ tri_out,
                              uint4* __restrict__ shade_out, uint32_t out_pieces) {
    constexpr uint32_t TP = sizeof(DTri) / 16, SP = sizeof(DTriShade) / 16;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = (uint32_t)(t / (TP + SP)), c = (uint32_t)(t % (TP + SP));
    if (i >= n) return;
    const uint32_t src = order[i];
    if (c < TP) tri_out[(uint64_t)i * out_pieces + c] = tri_in[(uint64_t)src * TP + c]; // out records sit out_pieces * 16 bytes apart
    else shade_out[(uint64_t)i * SP + (c - TP)] = shade_in[(uint64_t)src * SP + (c - TP)];
}
} // namespace

void launch_gather_tris(const DTri* tri_in, const DTriShade* shade_in, const uint32_t* order, uint32_t n, void* tri_out,
                        uint32_t tri_out_stride, DTriShade* shade_out, hipStream_t st) {
    static_assert(sizeof(DTri) % 16 == 0 && sizeof(DTriShade) % 16 == 0, "records are moved in 16-byte pieces");
    const uint64_t items = (uint64_t)n * ((sizeof(DTri) + sizeof(DTriShade)) / 16);
    k_gather_tris<<<(unsigned)((items + 255) / 256), 256, 0, st>>>(
        reinterpret_cast<const uint4*>(tri_in), reinterpret_cast<const uint4*>(shade_in), order, n,
        reinterpret_cast<uint4*>(tri_out), reinterpret_cast<uint4*>(shade_out), tri_out_stride / 16);
}

#define BVH_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            if (err) *err = std::string(#call) + ": " + hipGetErrorString(e_);          \
            return false;                                                               \
        }                                                                               \
    } while (0)

bool build_bvh_device(const PrimBox* h_boxes, const float box_origin[3], size_t n_, DeviceBVH& out, std::string* err) {
    static_assert(sizeof(PrimBox) == sizeof(FBox), "box layouts must agree");
    if (n_ < 2 || n_ >= ((size_t)1 << 28)) {
        if (err) *err = "device BVH build needs 2 <= triangles < 2^28";
        return false;
    }
    const uint32_t n = (uint32_t)n_;
    uint32_t M = 1;
    while (M < n) M <<= 1;
    const unsigned B = 256;
    auto blocks = [&](uint64_t items) { return (unsigned)std::max<uint64_t>(1, (items + B - 1) / B); };

    Scratch tmp;
    FBox *d_boxes = nullptr, *d_tree = nullptr, *d_root = nullptr;
    uint64_t *d_keys = nullptr, *d_keys2 = nullptr;
    uint32_t *d_vals = nullptr, *d_order = nullptr;
    BuildState* d_state = nullptr;
    Item *d_q0 = nullptr, *d_q1 = nullptr;
    FNodeD* d_fn = nullptr;
    BVH_HIP(tmp.alloc(&d_boxes, n));
    BVH_HIP(tmp.alloc(&d_tree, (size_t)2 * M));
    BVH_HIP(tmp.alloc(&d_root, 1));
    BVH_HIP(tmp.alloc(&d_keys, n));
    BVH_HIP(tmp.alloc(&d_keys2, n));
    BVH_HIP(tmp.alloc(&d_vals, n));
    BVH_HIP(tmp.alloc(&d_state, 1));
    BVH_HIP(tmp.alloc(&d_q0, n));
    BVH_HIP(tmp.alloc(&d_q1, n));
    BVH_HIP(tmp.alloc(&d_fn, n)); // a binary tree over n leaves has < n inner nodes
    BVH_HIP(hipMalloc(reinterpret_cast<void**>(&d_order), (size_t)n * sizeof(uint32_t)));
    struct OrderGuard { // d_order is handed to the caller only on success
        uint32_t*& p;
        bool keep = false;
        ~OrderGuard() {
            if (!keep && p) {
                (void)hipFree(p);
                p = nullptr;
            }
        }
    } order_guard{d_order};

    hipEvent_t ev[5];
    for (auto& e : ev) BVH_HIP(hipEventCreate(&e));
    struct EvGuard {
        hipEvent_t* e;
        ~EvGuard() {
            for (int i = 0; i < 5; ++i) (void)hipEventDestroy(e[i]);
        }
    } ev_guard{ev};

    BVH_HIP(hipMemcpy(d_boxes, h_boxes, (size_t)n * sizeof(FBox), hipMemcpyHostToDevice));
    hipStream_t st = nullptr;
    BVH_HIP(hipEventRecord(ev[0], st));
    k_init_state<<<1, 1, 0, st>>>(d_state);
    k_centroid_bounds<<<std::min(blocks(n), 2048u), B, 0, st>>>(d_boxes, n, d_state);
    k_morton<<<blocks(n), B, 0, st>>>(d_boxes, n, d_state, d_keys, d_vals);
    BVH_HIP(hipGetLastError());
    {
        size_t bytes = 0;
        BVH_HIP(rocprim::radix_sort_pairs(nullptr, bytes, d_keys, d_keys2, d_vals, d_order, n, 0u, (unsigned)kKeyBits, st));
        void* d_tmp = nullptr;
        BVH_HIP(tmp.alloc(reinterpret_cast<unsigned char**>(&d_tmp), bytes));
        BVH_HIP(rocprim::radix_sort_pairs(d_tmp, bytes, d_keys, d_keys2, d_vals, d_order, n, 0u, (unsigned)kKeyBits, st));
    }
    BVH_HIP(hipEventRecord(ev[1], st));
    k_tree_leaves<<<blocks(M), B, 0, st>>>(d_boxes, d_order, n, M, d_tree);
    for (uint32_t first = M >> 1; first >= 1; first >>= 1) k_tree_level<<<blocks(first), B, 0, st>>>(d_tree, first, first);
    k_root_box<<<1, 1, 0, st>>>(d_tree, d_root);
    BVH_HIP(hipGetLastError());
    BVH_HIP(hipEventRecord(ev[2], st));

    // level-synchronous splits; level L holds at most min(2^L, n) open nodes
    const Item root{0, n, kNoParent, 0};
    BVH_HIP(hipMemcpyAsync(d_q0, &root, sizeof(Item), hipMemcpyHostToDevice, st));
    {
        const uint32_t one = 1;
        BVH_HIP(hipMemcpyAsync(reinterpret_cast<char*>(d_state) + offsetof(BuildState, counts), &one, sizeof(one),
                               hipMemcpyHostToDevice, st));
    }
    Item *qin = d_q0, *qout = d_q1;
    for (int level = 0; level <= kMaxLevels; ++level) {
        const uint64_t cap = level >= 31 ? (uint64_t)n : std::min<uint64_t>((uint64_t)1 << level, n);
        k_split_level<<<blocks(cap), B, 0, st>>>(qin, qout, d_state, level, d_fn, d_tree, M, d_keys2);
        std::swap(qin, qout);
    }
    BVH_HIP(hipGetLastError());
    BVH_HIP(hipEventRecord(ev[3], st));

    BuildState hs;
    FBox root_box;
    BVH_HIP(hipMemcpy(&hs, d_state, sizeof(hs), hipMemcpyDeviceToHost)); // synchronises with the stream
    BVH_HIP(hipMemcpy(&root_box, d_root, sizeof(root_box), hipMemcpyDeviceToHost));
    if (hs.n_nodes == 0 || hs.n_nodes >= n || hs.counts[kMaxLevels + 1] != 0) {
        if (err) *err = "device BVH build: inconsistent node count";
        return false;
    }
    Grid g; // in the boxes' own coordinates (relative to box_origin, prim_boxes): the grid starts at 0 there
    {
        const float zero[3] = {0.f, 0.f, 0.f};
        quant_grid(zero, root_box.hi, false, g.origin, g.step);
    }
    DNode* d_nodes = nullptr;
    uint32_t n_out = 0, out_depth = 0;
    {
        // binary nodes of level L occupy [first[L], first[L] + counts[L+1]/2): heights bottom-up, then the collapse
        uint32_t first[kMaxLevels + 2], inner[kMaxLevels + 2];
        uint32_t acc = 0;
        for (int l = 0; l <= kMaxLevels; ++l) {
            first[l] = acc;
            inner[l] = hs.counts[l + 1] / 2;
            acc += inner[l];
        }
        if (acc != hs.n_nodes) {
            if (err) *err = "device BVH build: level ranges do not add up";
            return false;
        }
        uint8_t* d_h2 = nullptr;
        WideState* d_ws = nullptr;
        Open *d_o0 = nullptr, *d_o1 = nullptr;
        DNode* d_wide = nullptr;
        BVH_HIP(tmp.alloc(&d_h2, hs.n_nodes));
        BVH_HIP(tmp.alloc(&d_ws, 1));
        BVH_HIP(tmp.alloc(&d_o0, hs.n_nodes));
        BVH_HIP(tmp.alloc(&d_o1, hs.n_nodes));
        BVH_HIP(tmp.alloc(&d_wide, hs.n_nodes)); // a wide node replaces at least one binary node
        for (int l = kMaxLevels; l >= 0; --l)
            if (inner[l]) k_heights<<<blocks(inner[l]), B, 0, st>>>(d_fn, first[l], inner[l], d_h2);
        WideState ws0;
        std::memset(&ws0, 0, sizeof(ws0));
        ws0.n_wide = 1;
        ws0.counts[0] = 1;
        const Open root_open{0, 0, PRT_STACK_DEPTH};
        BVH_HIP(hipMemcpyAsync(d_ws, &ws0, sizeof(ws0), hipMemcpyHostToDevice, st));
        BVH_HIP(hipMemcpyAsync(d_o0, &root_open, sizeof(root_open), hipMemcpyHostToDevice, st));
        Open *oin = d_o0, *oout = d_o1;
        for (int level = 0; level <= kMaxLevels; ++level) {
            uint64_t cap = 1;
            for (int i = 0; i < level && cap < hs.n_nodes; ++i) cap *= 4;
            cap = std::min<uint64_t>(cap, hs.n_nodes);
            k_collapse_level<<<blocks(cap), B, 0, st>>>(oin, oout, d_ws, level, d_fn, d_h2, g, d_wide);
            std::swap(oin, oout);
        }
        BVH_HIP(hipGetLastError());
        WideState hw;
        BVH_HIP(hipMemcpy(&hw, d_ws, sizeof(hw), hipMemcpyDeviceToHost));
        if (hw.n_wide == 0 || hw.n_wide > hs.n_nodes || hw.counts[kMaxLevels + 1] != 0) {
            if (err) *err = "device BVH build: inconsistent wide node count";
            return false;
        }
        n_out = hw.n_wide;
        out_depth = hw.depth + 1;
        BVH_HIP(hipMalloc(reinterpret_cast<void**>(&d_nodes), std::max<size_t>((size_t)n_out * sizeof(DNode), 256)));
        hipError_t e = hipMemcpyAsync(d_nodes, d_wide, (size_t)n_out * sizeof(DNode), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(ev[4], st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            (void)hipFree(d_nodes);
            if (err) *err = std::string("device BVH build: ") + hipGetErrorString(e);
            return false;
        }
    }
    hs.n_nodes = n_out;
    hs.depth = out_depth;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ev[0], ev[1]); out.ms_sort = ms;
    (void)hipEventElapsedTime(&ms, ev[1], ev[2]); out.ms_tree = ms;
    (void)hipEventElapsedTime(&ms, ev[2], ev[3]); out.ms_split = ms;
    (void)hipEventElapsedTime(&ms, ev[0], ev[4]); out.ms_total = ms;

    out.d_nodes = d_nodes;
    out.d_order = d_order;
    order_guard.keep = true;
    out.n_nodes = hs.n_nodes;
    out.depth = hs.depth;
    for (int a = 0; a < 3; ++a) {
        out.grid_origin[a] = box_origin[a]; // what the kernels subtract from a ray's origin
        out.grid_step[a] = g.step[a];
    }
    float cs = 0.f;
    for (int a = 0; a < 3; ++a) cs = std::max(cs, std::max(std::fabs(root_box.lo[a]), std::fabs(root_box.hi[a])));
    cs = std::nextafter(cs, std::numeric_limits<float>::infinity());
    float gm = 0.f; // the dequantised coordinates can exceed the fp32 boxes by one grid step
    for (int a = 0; a < 3; ++a)
        gm = std::max(gm, std::fabs((float)((double)g.origin[a] + 65535.0 * (double)g.step[a])));
    cs = std::nextafter(std::max(cs, gm), std::numeric_limits<float>::infinity());
    out.coord_scale = cs;
    return true;
}

} // namespace prt

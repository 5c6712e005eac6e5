// bvh_build.cpp — host builder of the traversal BVH for the HIP kernels.
//
// The reference builds a pointer-based median-split tree with one triangle per leaf
// (Source/BVH.cpp:7-48).  Closest-hit results do not depend on the tree (only ties do), so this
// builder is free to produce what gfx950 traverses fastest: a binned-SAH BVH2, up to PRT_LEAF_MAX
// triangles per leaf, flattened pre-order into 64-byte nodes that carry BOTH children's boxes
// (one node fetch = two slab tests, two nodes per 128-byte L2 line), boxes stored as fp32 rounded
// outward (+ a small inflation that covers the rounding of the kernel's fma slab test) so the box
// test can only ever cull conservatively — every hit accept/reject is the fp64 triangle test.
// Depth is bounded so the per-lane LDS stack (PRT_STACK_DEPTH entries) cannot overflow.
//
// PRT_BVH_WIDTH 4: the same binary tree is then collapsed into 4-wide nodes (64 bytes: four child boxes + four
// refs) — a node absorbs the child with the largest box, repeatedly, until it has four children — so a ray makes
// about half as many dependent node fetches.  A traversal pushes up to three siblings per level, so the collapse
// carries a stack budget down the tree: a node with k children leaves budget - (k-1) entries to each child's
// subtree, and a child is only absorbed while every resulting child's binary subtree is no taller than what is
// left for it (a binary subtree of height h needs at most h entries).  The root starts with PRT_STACK_DEPTH.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>

#include "prt_host.h"

namespace prt {
namespace {

#ifndef PRT_SAH_BINS
#define PRT_SAH_BINS 64 // (round 4: 16 -> 64 bins: -1.3 % (cornell), -1.8 % (bathroom2) node visits on random rays, tools/sim_oct8.cpp; the build is not on the hot path)
#endif
constexpr int kBins = PRT_SAH_BINS;
constexpr int kMaxLevels = PRT_BVH2_LEVELS; // inner-node levels of the binary tree
#ifndef PRT_COST_TRI
#define PRT_COST_TRI 1.5f
#endif
constexpr float kCostTri = PRT_COST_TRI, kCostNode = 1.0f; // a triangle test is 8 loads + an fp64 division, a node visit 2 loads

struct PrimRef {
    float lo[3], hi[3];
    uint32_t idx;
};
struct FBox {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::numeric_limits<float>::infinity();
            hi[a] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const float* l, const float* h) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], l[a]);
            hi[a] = std::max(hi[a], h[a]);
        }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

inline float round_down(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafter(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float round_up(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafter(f, std::numeric_limits<float>::infinity());
    return f;
}

// builder-side node: fp32 outward-rounded child boxes, converted to the device format at the end
struct FNode {
    float c0x[2], c0y[2], c0z[2];
    float c1x[2], c1y[2], c1z[2];
    int32_t ref0, ref1;
};

struct Builder {
    std::vector<PrimRef> prims;
    std::vector<FNode>& nodes;
    uint32_t max_depth = 0;
    explicit Builder(std::vector<FNode>& n) : nodes(n) {}

    static int32_t leaf_ref(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

    void range_box(uint32_t s, uint32_t e, FBox& b) const {
        b.reset();
        for (uint32_t i = s; i < e; ++i) b.grow(prims[i].lo, prims[i].hi);
    }

    // Returns the ref of the subtree over prims[s,e) and its box.  `levels` = inner levels still allowed.
    int32_t build(uint32_t s, uint32_t e, int levels, uint32_t depth, FBox& box, bool force_inner) {
        const uint32_t count = e - s;
        range_box(s, e, box);
        max_depth = std::max(max_depth, depth);
        if (!force_inner && (count == 1 || levels == 0)) return leaf_ref(s, count);

        // centroid bounds
        float clo[3], chi[3];
        for (int a = 0; a < 3; ++a) {
            clo[a] = std::numeric_limits<float>::infinity();
            chi[a] = -clo[a];
        }
        for (uint32_t i = s; i < e; ++i)
            for (int a = 0; a < 3; ++a) {
                float c = 0.5f * (prims[i].lo[a] + prims[i].hi[a]);
                clo[a] = std::min(clo[a], c);
                chi[a] = std::max(chi[a], c);
            }

        // binned SAH over the three axes
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1, best_bin = -1;
        const float parent_area = std::max(box.half_area(), 1e-30f);
        for (int a = 0; a < 3; ++a) {
            const float ext = chi[a] - clo[a];
            if (!(ext > 0.f)) continue;
            FBox bb[kBins];
            uint32_t bc[kBins];
            for (int b = 0; b < kBins; ++b) {
                bb[b].reset();
                bc[b] = 0;
            }
            const float k = kBins * (1.f - 1e-6f) / ext;
            for (uint32_t i = s; i < e; ++i) {
                float c = 0.5f * (prims[i].lo[a] + prims[i].hi[a]);
                int b = std::min(kBins - 1, std::max(0, (int)((c - clo[a]) * k)));
                bb[b].grow(prims[i].lo, prims[i].hi);
                bc[b]++;
            }
            float right_area[kBins];
            uint32_t right_cnt[kBins];
            FBox acc;
            acc.reset();
            uint32_t n = 0;
            for (int b = kBins - 1; b > 0; --b) {
                if (bc[b]) acc.grow(bb[b].lo, bb[b].hi);
                n += bc[b];
                right_area[b] = n ? acc.half_area() : 0.f;
                right_cnt[b] = n;
            }
            acc.reset();
            n = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                if (bc[b]) acc.grow(bb[b].lo, bb[b].hi);
                n += bc[b];
                const uint32_t nr = right_cnt[b + 1];
                if (n == 0 || nr == 0) continue;
                const float cost = kCostNode + kCostTri * (acc.half_area() * n + right_area[b + 1] * nr) / parent_area;
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = a;
                    best_bin = b;
                }
            }
        }

        if (!force_inner && count <= PRT_LEAF_MAX && !(best_cost < kCostTri * count)) return leaf_ref(s, count);

        uint32_t mid = s;
        const uint64_t child_cap = (uint64_t)PRT_LEAF_MAX << (levels - 1);
        if (best_axis >= 0) {
            const int a = best_axis;
            const float k = kBins * (1.f - 1e-6f) / (chi[a] - clo[a]);
            const float c0 = clo[a];
            auto it = std::partition(prims.begin() + s, prims.begin() + e, [&](const PrimRef& p) {
                float c = 0.5f * (p.lo[a] + p.hi[a]);
                int b = std::min(kBins - 1, std::max(0, (int)((c - c0) * k)));
                return b <= best_bin;
            });
            mid = (uint32_t)(it - prims.begin());
        }
        if (mid == s || mid == e || std::max<uint64_t>(mid - s, e - mid) > child_cap) {
            // median split on the longest centroid axis (also the depth-bound fallback)
            int a = 0;
            if (chi[1] - clo[1] > chi[a] - clo[a]) a = 1;
            if (chi[2] - clo[2] > chi[a] - clo[a]) a = 2;
            mid = s + count / 2;
            std::nth_element(prims.begin() + s, prims.begin() + mid, prims.begin() + e,
                             [a](const PrimRef& x, const PrimRef& y) { return x.lo[a] + x.hi[a] < y.lo[a] + y.hi[a]; });
        }

        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        FBox b0, b1;
        const int32_t r0 = build(s, mid, levels - 1, depth + 1, b0, false);
        const int32_t r1 = build(mid, e, levels - 1, depth + 1, b1, false);
        FNode& n = nodes[me];
        n.c0x[0] = b0.lo[0]; n.c0x[1] = b0.hi[0];
        n.c0y[0] = b0.lo[1]; n.c0y[1] = b0.hi[1];
        n.c0z[0] = b0.lo[2]; n.c0z[1] = b0.hi[2];
        n.c1x[0] = b1.lo[0]; n.c1x[1] = b1.hi[0];
        n.c1y[0] = b1.lo[1]; n.c1y[1] = b1.hi[1];
        n.c1z[0] = b1.lo[2]; n.c1z[1] = b1.hi[2];
        n.ref0 = r0;
        n.ref1 = r1;
        return me;
    }
};

} // namespace

void prim_boxes(const std::vector<HostTri>& tris, std::vector<PrimBox>& out, float origin[3]) {
    // Boxes RELATIVE to `origin`, an fp32 point at or below every coordinate (it becomes the grid origin of the 16-bit
    // nodes; the kernels subtract it from the ray origin in the precision the ray comes in, prt_device.h slab_axis).  The
    // subtraction happens here in double: the fp32 boxes the builders bin, sort and quantise resolve the scene's EXTENT
    // to 2^-24, wherever the scene sits in the world — in absolute fp32 coordinates a unit scene at 1e6 had 16 distinct
    // positions per axis, its SAH tree cost 1.5x the node visits of the same scene at the origin.
    // A small absolute inflation on top of the outward rounding covers the fp64 roundings of the triangle test, which
    // scale with the magnitude of the coordinates (the kernel adds its own per-ray pad for the fp32 slab test).
    double scale = 1.0, lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (size_t i = 0; i < tris.size(); ++i)
        for (int a = 0; a < 3; ++a) {
            scale = std::max(scale, std::max(std::fabs(tris[i].lo[a]), std::fabs(tris[i].hi[a])));
            lo[a] = i ? std::min(lo[a], tris[i].lo[a]) : tris[i].lo[a];
            hi[a] = i ? std::max(hi[a], tris[i].hi[a]) : tris[i].hi[a];
        }
    const double extent = std::max(1.0, std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2])));
    const double delta = 1e-9 * extent + 256.0 * std::numeric_limits<double>::epsilon() * scale;
    for (int a = 0; a < 3; ++a) origin[a] = round_down(lo[a] - 2.0 * delta);
    out.resize(tris.size());
    for (size_t i = 0; i < tris.size(); ++i)
        for (int a = 0; a < 3; ++a) {
            out[i].lo[a] = round_down((tris[i].lo[a] - delta) - (double)origin[a]); // > 0: the origin lies 2 delta below
            out[i].hi[a] = round_up((tris[i].hi[a] + delta) - (double)origin[a]);
        }
}

void quant_grid(const float root_lo[3], const float root_hi[3], bool empty, float origin[3], float step[3]) {
    for (int a = 0; a < 3; ++a) {
        const double lo = empty ? 0.0 : (double)root_lo[a], hi = empty ? 1.0 : (double)root_hi[a];
        const float o = round_down(lo);
        float st = round_up((hi - (double)o) / 65535.0);
        if (!(st > 0.f)) st = std::numeric_limits<float>::min();
        while ((double)o + 65535.0 * (double)st < hi) st = std::nextafter(st, std::numeric_limits<float>::infinity());
        origin[a] = o;
        step[a] = st;
    }
}

// Structural check of a flattened tree (either builder, either width): every reference in range, every triangle
// in exactly one leaf, no traversal able to need more than PRT_STACK_DEPTH stack entries.  Used on device-built
// trees when PRT_VALIDATE_BVH is set (tests): a bad node index would be a GPU memory fault, not a wrong pixel.
bool validate_nodes(const DNode* nodes, size_t n_nodes, size_t n_tris, std::string* err) {
    auto bad = [&](const char* m) {
        if (err) *err = m;
        return false;
    };
    if (n_nodes == 0) return bad("no nodes");
    std::vector<uint8_t> seen(n_tris, 0);
    std::vector<int> need(n_nodes, -1);
    // iterative post-order over node indices (children may have any index in a device-built tree)
    std::vector<std::pair<uint32_t, int>> st;
    st.push_back({0u, 0});
    size_t visited = 0;
    while (!st.empty()) {
        auto [i, phase] = st.back();
        st.pop_back();
        int32_t refs[4];
        int nr = 0;
        for (int c = 0; c < 4; ++c)
            if (nodes[i].ref[c] != (int32_t)0x80000000) {
                if (c != nr) return bad("unused slot before a used one"); // the traversal only checks the refs of slots 2 and 3
                refs[nr++] = nodes[i].ref[c];
            }
        if (n_tris == 1 && nr == 2 && refs[1] == refs[0]) nr = 1; // the one-triangle root lists its leaf twice
        if (phase == 0) {
            if (++visited > n_nodes) return bad("cycle or shared node");
            if (nr < 1) return bad("node without children");
            if (PRT_BVH_WIDTH == 4 && nr < 2 && n_tris != 1) return bad("wide node with fewer than two children");
            st.push_back({i, 1});
            for (int c = 0; c < nr; ++c) {
                if (refs[c] >= 0) {
                    if ((size_t)refs[c] >= n_nodes) return bad("node index out of range");
                    if (need[refs[c]] != -1) return bad("node referenced twice");
                    need[refs[c]] = -2;
                    st.push_back({(uint32_t)refs[c], 0});
                } else {
                    const uint32_t enc = ~(uint32_t)refs[c], first = enc >> 3, cnt = (enc & 7u) + 1u;
                    if (cnt > PRT_LEAF_MAX || (size_t)first + cnt > n_tris) return bad("leaf range out of bounds");
                    for (uint32_t t = first; t < first + cnt; ++t)
                        if (seen[t]++) return bad("triangle in two leaves");
                }
            }
        } else {
            int worst = 0;
            for (int c = 0; c < nr; ++c)
                if (refs[c] >= 0) worst = std::max(worst, need[refs[c]]);
            need[i] = nr - 1 + worst;
        }
    }
    for (size_t t = 0; t < n_tris; ++t)
        if (!seen[t]) return bad("triangle in no leaf");
    if (need[0] > PRT_STACK_DEPTH) return bad("a traversal could overflow the stack");
    return true;
}

// Stack entries a traversal of this tree can need at most: a visit of a node with k children pushes up to k - 1 of
// them and descends into the remaining one, so need(node) = (k - 1) + max over inner children of need(child).  The
// builders bound this to PRT_STACK_DEPTH; most trees need far less, and a launch sizes its LDS stacks from it.
int tree_stack_need(const DNode* nodes, size_t n_nodes) {
    if (n_nodes == 0) return 0;
    std::vector<int> need(n_nodes, 0);
    std::vector<std::pair<uint32_t, int>> st;
    st.push_back({0u, 0});
    while (!st.empty()) {
        auto [i, phase] = st.back();
        st.pop_back();
        int32_t refs[4];
        int nr = 0;
        for (int c = 0; c < 4; ++c)
            if (nodes[i].ref[c] != (int32_t)0x80000000) refs[nr++] = nodes[i].ref[c];
        if (phase == 0) {
            st.push_back({i, 1});
            for (int c = 0; c < nr; ++c)
                if (refs[c] >= 0 && (size_t)refs[c] < n_nodes) st.push_back({(uint32_t)refs[c], 0});
        } else {
            int worst = 0;
            for (int c = 0; c < nr; ++c)
                if (refs[c] >= 0 && (size_t)refs[c] < n_nodes) worst = std::max(worst, need[refs[c]]);
            need[i] = std::max(0, nr - 1) + worst;
        }
    }
    return need[0];
}

bool build_bvh(const std::vector<HostTri>& tris, BuiltBVH& out, std::string* err) {
    out.nodes.clear();
    out.order.clear();
    out.depth = 0;
    const size_t n = tris.size();
    if (n >= ((size_t)1 << 28)) {
        if (err) *err = "too many triangles for the 28-bit leaf encoding";
        return false;
    }
    std::vector<FNode> fn;
    if (n == 0) { // traversal short-circuits on n_tris == 0; keep a well-formed root anyway
        FNode z;
        std::memset(&z, 0, sizeof(z));
        z.ref0 = z.ref1 = ~0;
        fn.push_back(z);
    }
    Builder b(fn);
    b.prims.resize(n);
    float box_origin[3] = {0.f, 0.f, 0.f}; // every fp32 box below is relative to this point (prim_boxes)
    {
        std::vector<PrimBox> pb;
        prim_boxes(tris, pb, box_origin);
        for (size_t i = 0; i < n; ++i) {
            std::memcpy(b.prims[i].lo, pb[i].lo, sizeof(float) * 3);
            std::memcpy(b.prims[i].hi, pb[i].hi, sizeof(float) * 3);
            b.prims[i].idx = (uint32_t)i;
        }
    }
    FBox root;
    root.reset();
    if (n == 1) {
        // one triangle: the root tests it through both children (like the reference's span-1 node, BVH.cpp:21-23)
        b.range_box(0, 1, root);
        FNode r;
        std::memset(&r, 0, sizeof(r));
        r.c0x[0] = r.c1x[0] = root.lo[0]; r.c0x[1] = r.c1x[1] = root.hi[0];
        r.c0y[0] = r.c1y[0] = root.lo[1]; r.c0y[1] = r.c1y[1] = root.hi[1];
        r.c0z[0] = r.c1z[0] = root.lo[2]; r.c0z[1] = r.c1z[1] = root.hi[2];
        r.ref0 = r.ref1 = Builder::leaf_ref(0, 1);
        fn.push_back(r);
    } else if (n > 1) {
        fn.reserve(n);
        b.build(0, (uint32_t)n, kMaxLevels, 0, root, true);
    }
    out.depth = b.max_depth;
    out.binary.clear();
    if (out.keep_binary) { // tools/sim_oct8.cpp
        out.binary.resize(fn.size());
        for (size_t i = 0; i < fn.size(); ++i) {
            BuiltBVH::BinNode& o = out.binary[i];
            const float* src[2][3] = {{fn[i].c0x, fn[i].c0y, fn[i].c0z}, {fn[i].c1x, fn[i].c1y, fn[i].c1z}};
            for (int c = 0; c < 2; ++c)
                for (int a = 0; a < 3; ++a) { // (back in world coordinates: the simulator's rays are)
                    o.lo[c][a] = round_down((double)box_origin[a] + (double)src[c][a][0]);
                    o.hi[c][a] = round_up((double)box_origin[a] + (double)src[c][a][1]);
                }
            o.ref[0] = fn[i].ref0;
            o.ref[1] = fn[i].ref1;
        }
    }
    float cs = 0.f;
    for (const PrimRef& pr : b.prims)
        for (int a = 0; a < 3; ++a) cs = std::max(cs, std::max(std::fabs(pr.lo[a]), std::fabs(pr.hi[a])));
    out.coord_scale = n ? std::nextafter(cs, std::numeric_limits<float>::infinity()) : 1.0f;
    out.order.resize(n);
    for (size_t i = 0; i < n; ++i) out.order[i] = b.prims[i].idx;

    // quantisation grid over the root box: coordinate(q) = g0 + q * gs, evaluated in double here; the
    // kernel's float evaluation error is covered by its per-ray pad.  lo rounds down, hi rounds up.  The boxes are
    // relative to box_origin, the grid starts there (g0 = 0 in the builder's coordinates), and box_origin is what the
    // kernels get as the grid origin.
    double g0[3], gs[3];
    {
        const float zero[3] = {0.f, 0.f, 0.f};
        float rel_origin[3];
        quant_grid(zero, root.hi, n == 0, rel_origin, out.grid_step);
    }
    for (int a = 0; a < 3; ++a) {
        out.grid_origin[a] = box_origin[a];
        g0[a] = 0.0;
        gs[a] = out.grid_step[a];
    }
    auto qlo = [&](float v, int a) -> uint16_t {
        double q = std::floor(((double)v - g0[a]) / gs[a]);
        q = std::min(65535.0, std::max(0.0, q));
        while (q > 0 && g0[a] + q * gs[a] > (double)v) q -= 1;
        return (uint16_t)q;
    };
    auto qhi = [&](float v, int a) -> uint16_t {
        double q = std::ceil(((double)v - g0[a]) / gs[a]);
        q = std::min(65535.0, std::max(0.0, q));
        while (q < 65535 && g0[a] + q * gs[a] < (double)v) q += 1;
        return (uint16_t)q;
    };
    {
        // heights of the binary subtrees (children follow their parent in the pre-order array)
        std::vector<uint8_t> h2(fn.size(), 1);
        for (size_t i = fn.size(); i-- > 0;) {
            const int a = fn[i].ref0 >= 0 ? h2[fn[i].ref0] : 0, c = fn[i].ref1 >= 0 ? h2[fn[i].ref1] : 0;
            h2[i] = (uint8_t)(1 + std::max(a, c));
        }
        struct Kid {
            int32_t ref; // ref in the BINARY tree (inner index or leaf ref)
            float lo[3], hi[3];
        };
        auto kid_of = [&](const FNode& f, int side) {
            Kid k;
            k.ref = side ? f.ref1 : f.ref0;
            const float* x = side ? f.c1x : f.c0x;
            const float* y = side ? f.c1y : f.c0y;
            const float* z = side ? f.c1z : f.c0z;
            k.lo[0] = x[0]; k.hi[0] = x[1]; k.lo[1] = y[0]; k.hi[1] = y[1]; k.lo[2] = z[0]; k.hi[2] = z[1];
            return k;
        };
        auto height = [&](int32_t ref) { return ref >= 0 ? (int)h2[ref] : 0; };
        auto area = [](const Kid& k) {
            const float dx = k.hi[0] - k.lo[0], dy = k.hi[1] - k.lo[1], dz = k.hi[2] - k.lo[2];
            return dx * dy + dy * dz + dz * dx;
        };
        struct Open {
            int32_t bin;   // binary node to collapse
            uint32_t slot; // index of the wide node it becomes
            int budget;    // stack entries its subtree may use
            uint32_t depth;
        };
        // Collapses the binary tree into `nodes` such that no traversal can need more than `stack_budget` entries.
        auto collapse = [&](int stack_budget, std::vector<DNode>& nodes) -> uint32_t {
        std::vector<Open> todo;
        nodes.clear();
        nodes.emplace_back();
        todo.push_back({0, 0, stack_budget, 0});
        uint32_t wide_depth = 0;
        while (!todo.empty()) {
            const Open o = todo.back();
            todo.pop_back();
            wide_depth = std::max(wide_depth, o.depth);
            Kid kids[4];
            int nk = 0;
            kids[nk++] = kid_of(fn[o.bin], 0);
            kids[nk++] = kid_of(fn[o.bin], 1); // every wide node has >= 2 children in slots 0, 1 (the traversal relies on it): the
                                               // one-triangle root keeps both copies of its leaf, like the reference's span-1 node
            while (nk < 4) {
                int best = -1;
                float best_area = -1.f;
                for (int i = 0; i < nk; ++i) {
                    if (kids[i].ref < 0) continue;
                    // absorbing kids[i] makes nk+1 children; each of them keeps o.budget - nk entries
                    const int left = o.budget - nk;
                    bool fits = height(fn[kids[i].ref].ref0) <= left && height(fn[kids[i].ref].ref1) <= left;
                    for (int j = 0; j < nk && fits; ++j)
                        if (j != i && height(kids[j].ref) > left) fits = false;
                    if (fits && area(kids[i]) > best_area) {
                        best_area = area(kids[i]);
                        best = i;
                    }
                }
                if (best < 0) break;
                const FNode& f = fn[kids[best].ref];
                kids[best] = kid_of(f, 0);
                kids[nk++] = kid_of(f, 1);
            }
            // the inner children get consecutive wide nodes (two per 128-byte line)
            DNode d;
            for (int i = 0; i < 4; ++i) {
                d.bx[i] = d.by[i] = d.bz[i] = 0x0000ffffu; // lo = 0xffff, hi = 0: never hit
                d.ref[i] = (int32_t)0x80000000;
            }
            for (int i = 0; i < nk; ++i) {
                d.bx[i] = (uint32_t)qlo(kids[i].lo[0], 0) | ((uint32_t)qhi(kids[i].hi[0], 0) << 16);
                d.by[i] = (uint32_t)qlo(kids[i].lo[1], 1) | ((uint32_t)qhi(kids[i].hi[1], 1) << 16);
                d.bz[i] = (uint32_t)qlo(kids[i].lo[2], 2) | ((uint32_t)qhi(kids[i].hi[2], 2) << 16);
                if (kids[i].ref < 0) d.ref[i] = kids[i].ref;
                else {
                    d.ref[i] = (int32_t)nodes.size();
                    nodes.emplace_back();
                }
            }
            for (int i = nk - 1; i >= 0; --i) // depth-first, first child next
                if (kids[i].ref >= 0) todo.push_back({kids[i].ref, (uint32_t)d.ref[i], o.budget - (nk - 1), o.depth + 1});
            nodes[o.slot] = d;
        }
        return wide_depth + 1;
        };
        // Cost-optimal collapse (round 4; Ylitie, Karras, Laine 2017, section 3): a visit costs the kernel the same whatever the
        // node holds, and a node's surface area is the probability of visiting it, so the tree to want is the one with the
        // least SUMMED AREA OF WIDE NODES.  c(n, i) = least cost of turning the binary subtree n into a forest of at most i
        // wide-tree roots: c(n, 1) = area(n) + best split of four roots over n's two subtrees (n becomes a wide node),
        // c(n, i) = min(c(n, i - 1), best split of i roots over the two subtrees).  The builder's leaves stay leaves.  Against
        // the greedy largest-box rule: 20-28 % fewer nodes and, on random rays, -0.9 % (cornell) / -1.9 % (bathroom2) / -9.7 %
        // (veach-mis) node visits (tools/sim_oct8.cpp).  The greedy collapse with its stack budget remains the rule of the 32-entry
        // tree, of trees beyond a million binary nodes and of the GPU builder.
        // The stack bound enters the programme as a LEVEL: a wide node at level L of the wide tree is entered with at most
        // 3 L entries on the stack (every ancestor pushed at most three siblings), so a tree of at most (entries - 1) / 3 levels can
        // never need more than `entries`: c(n, i, L) = least cost of a forest of at most i roots AT LEVEL L over
        // the binary subtree n; a wide node at the last level may only have leaves below it.  (Conservative by up to two
        // entries per level against the exact k - 1; a deep tree — bathroom2: 30 binary levels — still gets 13 wide ones.)
        auto collapse_optimal = [&](int stack_entries, std::vector<DNode>& nodes) -> uint32_t {
            const size_t nb = fn.size();
            constexpr int W = 4;
            const int L = (stack_entries - 1) / 3; // level l < L: 3 l + 3 <= stack_entries
            const float inf = std::numeric_limits<float>::infinity();
            std::vector<float> c(nb * L * (W + 1), 0.f);
            std::vector<int8_t> split(nb * L * (W + 1), 0); // roots given to the left subtree; 0 = "same as with one root fewer"
            auto at = [&](size_t n_, int lev, int i) { return (n_ * L + (size_t)lev) * (W + 1) + (size_t)i; };
            auto cost = [&](int32_t ref, int lev, int i) -> float {
                if (ref < 0) return 0.f;           // a leaf costs no node visit, at any level
                if (lev >= L) return inf;          // no wide node below the last level
                return c[at((size_t)ref, lev, std::min(i, W))];
            };
            for (size_t i = nb; i-- > 0;) { // children follow their parent in the pre-order array
                const Kid k0 = kid_of(fn[i], 0), k1 = kid_of(fn[i], 1);
                Kid u = k0;
                for (int a = 0; a < 3; ++a) {
                    u.lo[a] = std::min(k0.lo[a], k1.lo[a]);
                    u.hi[a] = std::max(k0.hi[a], k1.hi[a]);
                }
                const float au = area(u);
                for (int lev = L - 1; lev >= 0; --lev) {
                    auto distribute = [&](int j, int at_level, int8_t& best_a) {
                        float best = inf;
                        for (int a = 1; a < j; ++a) {
                            const float v = cost(k0.ref, at_level, a) + cost(k1.ref, at_level, j - a);
                            if (v < best) {
                                best = v;
                                best_a = (int8_t)a;
                            }
                        }
                        return best;
                    };
                    int8_t a1 = 1;
                    c[at(i, lev, 1)] = au + distribute(W, lev + 1, a1); // this binary node becomes a wide node at `lev`
                    split[at(i, lev, 1)] = a1;
                    for (int j = 2; j <= W; ++j) {
                        int8_t aj = 1;
                        const float dcost = distribute(j, lev, aj); // j roots at this level, over the two subtrees
                        if (dcost < c[at(i, lev, j - 1)]) {
                            c[at(i, lev, j)] = dcost;
                            split[at(i, lev, j)] = aj;
                        } else {
                            c[at(i, lev, j)] = c[at(i, lev, j - 1)];
                            split[at(i, lev, j)] = 0;
                        }
                    }
                }
            }
            if (!(c[at(0, 0, 1)] < inf)) return 0; // (cannot happen for a binary tree of PRT_BVH2_LEVELS levels; the caller falls back)
            struct Emit { int32_t bin; uint32_t slot; uint32_t depth; };
            std::vector<Emit> todo;
            nodes.clear();
            nodes.emplace_back();
            todo.push_back({0, 0, 0});
            uint32_t wide_depth = 0;
            Kid kids[4];
            int nk = 0;
            // the roots, at level `lev`, of the best forest of at most i trees over the binary subtree `k.ref` (box k)
            auto forest = [&](auto&& self, const Kid& k, int lev, int i) -> void {
                if (k.ref >= 0) {
                    i = std::min(i, W);
                    while (i > 1 && split[at((size_t)k.ref, lev, i)] == 0) --i;
                    if (i > 1) {
                        const int a = split[at((size_t)k.ref, lev, i)];
                        self(self, kid_of(fn[k.ref], 0), lev, a);
                        self(self, kid_of(fn[k.ref], 1), lev, i - a);
                        return;
                    }
                }
                kids[nk++] = k; // a leaf, or a wide node of its own at `lev`
            };
            while (!todo.empty()) {
                const Emit o = todo.back();
                todo.pop_back();
                wide_depth = std::max(wide_depth, o.depth);
                nk = 0;
                if (n == 1) { // the one-triangle root keeps both copies of its leaf (slots 0 and 1 are always used)
                    kids[nk++] = kid_of(fn[o.bin], 0);
                    kids[nk++] = kid_of(fn[o.bin], 1);
                } else {
                    const int a = split[at((size_t)o.bin, (int)o.depth, 1)];
                    forest(forest, kid_of(fn[o.bin], 0), (int)o.depth + 1, a);
                    forest(forest, kid_of(fn[o.bin], 1), (int)o.depth + 1, W - a);
                }
                DNode d;
                for (int i = 0; i < 4; ++i) {
                    d.bx[i] = d.by[i] = d.bz[i] = 0x0000ffffu;
                    d.ref[i] = (int32_t)0x80000000;
                }
                for (int i = 0; i < nk; ++i) {
                    d.bx[i] = (uint32_t)qlo(kids[i].lo[0], 0) | ((uint32_t)qhi(kids[i].hi[0], 0) << 16);
                    d.by[i] = (uint32_t)qlo(kids[i].lo[1], 1) | ((uint32_t)qhi(kids[i].hi[1], 1) << 16);
                    d.bz[i] = (uint32_t)qlo(kids[i].lo[2], 2) | ((uint32_t)qhi(kids[i].hi[2], 2) << 16);
                    if (kids[i].ref < 0) d.ref[i] = kids[i].ref;
                    else {
                        d.ref[i] = (int32_t)nodes.size();
                        nodes.emplace_back();
                    }
                }
                for (int i = nk - 1; i >= 0; --i)
                    if (kids[i].ref >= 0) todo.push_back({kids[i].ref, (uint32_t)d.ref[i], o.depth + 1});
                nodes[o.slot] = d;
            }
            return wide_depth + 1;
        };
        // (the programme's table is 13 x 5 floats + bytes per binary node, 340 MB of transient host memory at a million nodes:
        // trees beyond that keep the greedy rule, and so does a host that cannot spare the table)
        bool dp = fn.size() <= ((size_t)1 << 20);
        auto try_optimal = [&](int entries, std::vector<DNode>& nodes) -> uint32_t {
            if (!dp) return 0;
            try {
                return collapse_optimal(entries, nodes);
            } catch (const std::bad_alloc&) {
                dp = false;
                return 0;
            }
        };
        out.depth = try_optimal(PRT_STACK_DEPTH, out.nodes);
        if (out.depth == 0 || tree_stack_need(out.nodes.data(), out.nodes.size()) > PRT_STACK_DEPTH) out.depth = collapse(PRT_STACK_DEPTH, out.nodes);
        out.stack_need = tree_stack_need(out.nodes.data(), out.nodes.size());
        out.nodes_shallow.clear();
        if (out.stack_need > PRT_STACK_SHALLOW) {
            const uint32_t ok = try_optimal(PRT_STACK_SHALLOW, out.nodes_shallow);
            if (ok == 0 || tree_stack_need(out.nodes_shallow.data(), out.nodes_shallow.size()) > PRT_STACK_SHALLOW) collapse(PRT_STACK_SHALLOW, out.nodes_shallow);
        }
    }
    float gm = 0.f; // the dequantised coordinates can exceed the fp32 boxes by one grid step
    for (int a = 0; a < 3; ++a)
        gm = std::max(gm, std::fabs((float)(g0[a] + 65535.0 * gs[a])));
    out.coord_scale = std::nextafter(std::max(out.coord_scale, gm), std::numeric_limits<float>::infinity());
    return true;
}

} // namespace prt

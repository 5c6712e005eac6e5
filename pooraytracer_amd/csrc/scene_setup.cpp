// scene_setup.cpp — host-side scene preparation: per-triangle precompute, material table, the
// reference-ordered light tree and the camera frame.  Pure host C++ (no device code); runs once per
// scene / per frame, never per ray.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "prt_host.h"

namespace prt {
namespace {

struct V {
    double x, y, z;
};
inline V sub(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V add(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V scale(V a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V cross(V a, V b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V unit(V a) { return scale(a, 1.0 / std::sqrt(dot(a, a))); } // glm::normalize
inline bool has_nan(V a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
inline void store(double* p, V a) {
    p[0] = a.x;
    p[1] = a.y;
    p[2] = a.z;
}
inline V load(const double* p) { return {p[0], p[1], p[2]}; }

// AABB(a,b) + PadToMinimus per axis (AABB.cpp:16-22,76-82), then the union of the two edge boxes
// (Triangle.cpp:94-99).  The pad is applied to each edge box before the union, as in the reference.
inline void edge_interval(double a, double b, double& lo, double& hi) {
    lo = (a <= b) ? a : b;
    hi = (a <= b) ? b : a;
    if (hi - lo < 0.0001) {
        lo -= 0.0001 / 2.;
        hi += 0.0001 / 2.;
    }
}

} // namespace

// Triangle::Triangle (Triangle.cpp:11-53) for one triangle whose vertices p[], vertex normals vn[] and T.uv are given.
static void setup_triangle_geometry(HostTri& T, const V p[3], const V vn[3]) {
    for (int k = 0; k < 3; ++k) store(T.v[k], p[k]);
    V e0 = sub(p[1], p[0]), e1 = sub(p[2], p[0]);
    V n = cross(e0, e1);
    V nn = unit(n);
    if (has_nan(nn)) { // degenerate face: vertex-normal fallback, then +z (Triangle.cpp:21-29)
        nn = unit(add(add(vn[0], vn[1]), vn[2]));
        if (has_nan(nn)) nn = V{0.0, 0.0, 1.0};
    }
    // tangent from the UV deltas (Triangle.cpp:31-37)
    double du0 = T.uv[1][0] - T.uv[0][0], dv0 = T.uv[1][1] - T.uv[0][1];
    double du1 = T.uv[2][0] - T.uv[0][0], dv1 = T.uv[2][1] - T.uv[0][1];
    double f = 1.0 / (du0 * dv1 - du1 * dv0);
    V tg{f * (dv1 * e0.x - dv0 * e1.x), f * (dv1 * e0.y - dv0 * e1.y), f * (dv1 * e0.z - dv0 * e1.z)};
    tg = unit(tg);
    if (has_nan(tg)) { // Triangle.cpp:39-46 (0.9f: float literal)
        V helper = (std::fabs(nn.x) < (double)0.9f) ? V{1, 0, 0} : V{0, 1, 0};
        tg = unit(cross(nn, helper));
    }
    store(T.e0, e0);
    store(T.e1, e1);
    store(T.normal, nn);
    store(T.tangent, tg);
    T.area = std::sqrt(dot(n, n)) * 0.5;
    T.D = dot(nn, p[0]);
    double nn2 = dot(n, n);
    T.w[0] = n.x / nn2;
    T.w[1] = n.y / nn2;
    T.w[2] = n.z / nn2;
    for (int a = 0; a < 3; ++a) {
        double l0, h0, l1, h1;
        edge_interval(T.v[0][a], T.v[1][a], l0, h0);
        edge_interval(T.v[0][a], T.v[2][a], l1, h1);
        T.lo[a] = l0 <= l1 ? l0 : l1;
        T.hi[a] = h0 >= h1 ? h0 : h1;
    }
}

void setup_triangles(const PrtSceneDesc& d, std::vector<HostTri>& out) {
    out.resize(d.n_tris);
    for (uint32_t m = 0; m < d.n_meshes; ++m) {
        for (uint64_t t = d.mesh_first_tri[m]; t < d.mesh_first_tri[m + 1]; ++t) {
            HostTri& T = out[t];
            V p[3], vn[3];
            for (int k = 0; k < 3; ++k) {
                p[k] = load(d.vertices + t * 9 + k * 3);
                vn[k] = d.normals ? load(d.normals + t * 9 + k * 3) : V{0, 0, 0};
                T.uv[k][0] = d.texcoords ? d.texcoords[t * 6 + k * 2] : 0.0;
                T.uv[k][1] = d.texcoords ? d.texcoords[t * 6 + k * 2 + 1] : 0.0;
            }
            T.material = d.mesh_material[m];
            T.prim = (int32_t)t;
            setup_triangle_geometry(T, p, vn);
        }
    }
}

// New vertex positions for an existing triangle set (same meshes, materials and texture coordinates).
void update_triangles(const double* vertices, const double* normals, std::vector<HostTri>& tris) {
    for (size_t t = 0; t < tris.size(); ++t) {
        V p[3], vn[3];
        for (int k = 0; k < 3; ++k) {
            p[k] = load(vertices + t * 9 + k * 3);
            vn[k] = normals ? load(normals + t * 9 + k * 3) : V{0, 0, 0};
        }
        setup_triangle_geometry(tris[t], p, vn);
    }
}

void setup_materials(const PrtSceneDesc& d, std::vector<DMaterial>& out) {
    out.resize(d.n_materials);
    for (uint32_t i = 0; i < d.n_materials; ++i) {
        const PrtMaterial& s = d.materials[i];
        DMaterial& m = out[i];
        std::memset(&m, 0, sizeof(m));
        m.type = s.type;
        m.texture = s.texture;
        for (int c = 0; c < 3; ++c) {
            m.kd[c] = s.kd[c];
            m.ks[c] = s.ks[c];
            m.eta[c] = s.eta[c];
            m.k[c] = s.k[c];
        }
        m.ns = s.ns;
        m.inv_ns1 = 1.0 / (s.ns + 1.0);
        m.spec_scale = (s.ns + 2.0) / (s.ns + 1.0);
        // SetProbabilitiesByNs, Material.h:318-327
        if (s.ns <= 9.) {
            m.pkd = 1.0;
            m.pks = 0.0;
        } else {
            m.pkd = 0.6;
            m.pks = 0.4;
        }
        m.alpha_x = s.alpha_x;
        m.alpha_y = s.alpha_y;
        m.has_emission = (s.type == PRT_MAT_DIFFUSE_LIGHT || s.type == PRT_MAT_DEBUG) ? 1 : 0; // Material.h:168,527
        for (int c = 0; c < 3; ++c)
            m.emission[c] = s.type == PRT_MAT_DIFFUSE_LIGHT ? s.emission[c] : (s.type == PRT_MAT_DEBUG ? s.kd[c] : 0.0);
        // SkipLightSampling: Material.h:73 (false), :328 Phong Ns>1, :365 mirror, :539 empty
        m.skip_light_sampling =
            (s.type == PRT_MAT_MIRROR || s.type == PRT_MAT_EMPTY || (s.type == PRT_MAT_PHONG && s.ns > 1.)) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Light tree.  NEE picks a light triangle by descending the reference's lights BVH with an area
// CDF (BVH.cpp:86-100); which triangle a given `p` lands on therefore depends on that tree's shape:
// a two-level median-split build (main.cpp:36-45 -> BVH.cpp:7-48) whose std::sort reorders each
// mesh's triangle list in place.  For an emissive mesh the list is sorted once by the world build
// (main.cpp:39) and then again by the lights build (main.cpp:41); both passes are replayed here so
// the leaf order is the reference's.  Only emissive meshes are touched (tiny), never the world.
namespace {

struct Obj { // a Hittable handle: triangle (node < 0) or tree node
    int32_t tri;  // index into HostTri when node < 0
    int32_t node; // index into TNode otherwise
};
struct TNode {
    double lo[3], hi[3];
    double area;
    Obj left, right;
    bool single; // span-1 node: left == right
};

struct RefBuilder {
    const std::vector<HostTri>& tris;
    std::vector<TNode> nodes;
    explicit RefBuilder(const std::vector<HostTri>& t) : tris(t) {}

    const double* lo(const Obj& o) const { return o.node < 0 ? tris[o.tri].lo : nodes[o.node].lo; }
    const double* hi(const Obj& o) const { return o.node < 0 ? tris[o.tri].hi : nodes[o.node].hi; }
    double area(const Obj& o) const { return o.node < 0 ? tris[o.tri].area : nodes[o.node].area; }

    int32_t build(std::vector<Obj>& objs, size_t start, size_t end) { // BVH.cpp:7-48
        TNode n;
        const double inf = std::numeric_limits<double>::infinity();
        for (int a = 0; a < 3; ++a) {
            n.lo[a] = inf;
            n.hi[a] = -inf;
        }
        for (size_t i = start; i < end; ++i)
            for (int a = 0; a < 3; ++a) {
                const double l = lo(objs[i])[a], h = hi(objs[i])[a];
                n.lo[a] = n.lo[a] <= l ? n.lo[a] : l; // Interval(a,b) union, Interval.h:14-17
                n.hi[a] = n.hi[a] >= h ? n.hi[a] : h;
            }
        const double lx = n.hi[0] - n.lo[0], ly = n.hi[1] - n.lo[1], lz = n.hi[2] - n.lo[2];
        const int axis = (lx > ly) ? (lx > lz ? 0 : 2) : (ly > lz ? 1 : 2); // AABB.cpp:66-74
        const size_t span = end - start;
        n.single = false;
        if (span == 1) {
            n.left = n.right = objs[start];
            n.area = area(objs[start]);
            n.single = true;
        } else if (span == 2) {
            n.left = objs[start];
            n.right = objs[start + 1];
            n.area = area(objs[start]) + area(objs[start + 1]);
        } else {
            std::sort(objs.begin() + start, objs.begin() + end,
                      [&](const Obj& a, const Obj& b) { return lo(a)[axis] < lo(b)[axis]; });
            const size_t mid = start + span / 2;
            const int32_t l = build(objs, start, mid);
            const int32_t r = build(objs, mid, end);
            n.left = Obj{-1, l};
            n.right = Obj{-1, r};
            n.area = nodes[l].area + nodes[r].area;
        }
        nodes.push_back(n);
        return (int32_t)nodes.size() - 1;
    }
};

int32_t flatten_light(const RefBuilder& rb, const Obj& o, LightTree& out, const std::vector<HostTri>& tris) {
    if (o.node < 0) {
        const HostTri& T = tris[o.tri];
        DLightTri lt;
        std::memset(&lt, 0, sizeof(lt));
        std::memcpy(lt.v0, T.v[0], 24);
        std::memcpy(lt.v1, T.v[1], 24);
        std::memcpy(lt.v2, T.v[2], 24);
        std::memcpy(lt.n, T.normal, 24);
        lt.area = T.area;
        lt.material = T.material;
        lt.prim = T.prim;
        out.tris.push_back(lt);
        return ~(int32_t)(out.tris.size() - 1);
    }
    const TNode& n = rb.nodes[o.node];
    // A span-1 node (left == right, BVH.cpp:21-23) over a TRIANGLE: either branch samples that triangle.  Over a NODE
    // (the top-level node of a lights list with one mesh, main.cpp:45) the two branches differ: TraverseSample goes right
    // when the float p is not below the child's area — sqrt(xi) * area rounds UP to the whole area for xi > 1 - 2^-24 —
    // and descends the same child with p - area = 0, i.e. wraps around to the FIRST triangle of the CDF (BVH.cpp:93-98).
    // The node is kept, with both refs on the one flattened child.
    if (n.single && n.left.node < 0) return flatten_light(rb, n.left, out, tris);
    const int32_t idx = (int32_t)out.nodes.size();
    out.nodes.push_back(DLightNode{rb.area(n.left), 0, 0});
    const int32_t l = flatten_light(rb, n.left, out, tris);
    const int32_t r = n.single ? l : flatten_light(rb, n.right, out, tris);
    out.nodes[idx].left = l;
    out.nodes[idx].right = r;
    return idx;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// O(1) light pick: threshold + bucket tables for the subtrees on which TraverseSample is monotone (prt_types.h, DLightTable).
namespace {
inline float bits_float(uint32_t b) {
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}
inline uint32_t float_bits(float f) {
    uint32_t b;
    std::memcpy(&b, &f, 4);
    return b;
}
// BVHNode::TraverseSample (BVH.cpp:86-100) on the flattened nodes, as sample_lights executes it: the comparison and the
// subtraction in double, p truncated to float at every level (the parameter's type, BVH.h:32).  Returns the leaf's CDF index.
int32_t descend(const std::vector<DLightNode>& nodes, int32_t node, float p) {
    while (node >= 0) {
        const DLightNode& ln = nodes[node];
        if ((double)p < ln.left_area) node = ln.left;
        else {
            p = (float)((double)p - ln.left_area);
            node = ln.right;
        }
    }
    return ~node;
}
inline uint32_t bucket_of(float p, float inv_w, uint32_t n_bkt) { // the kernel's expression, operation for operation
    const float x = std::fmin(p * inv_w, (float)(n_bkt - 1));
    return (uint32_t)x;
}
// smallest float bit pattern b in [0, 0x7f7fffff] (positive floats order like their patterns) with pred(bits_float(b)); 0x7f800000 if none
template <typename Pred>
uint32_t first_pattern(Pred pred) {
    if (!pred(bits_float(0x7f7fffffu))) return 0x7f800000u;
    uint32_t lo = 0, hi = 0x7f7fffffu; // invariant: pred(hi)
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (pred(bits_float(mid))) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}
int32_t table_pick(const uint32_t* tab, uint32_t table, float p) { // host twin of the kernel's table lookup
    DLightTable t;
    std::memcpy(&t, tab + 8 * (size_t)table, sizeof(t));
    const uint32_t k = bucket_of(p, t.inv_w, t.n_bkt);
    uint32_t i = tab[t.bkt_off + 2 * k];
    float next = bits_float(tab[t.bkt_off + 2 * k + 1]);
    while (next <= p) {
        ++i;
        next = bits_float(tab[t.thr_off + i + 1]);
    }
    return (int32_t)(t.first + i);
}
} // namespace

int32_t light_pick_full_tree(const LightTree& lt, float p) { return descend(lt.full_nodes, lt.full_root, p); }
int32_t light_pick(const LightTree& lt, float p) {
    int32_t node = lt.root;
    while (node >= 0) {
        if (node & PRT_LIGHT_TABLE_BIT) return table_pick(lt.tab.data(), (uint32_t)node & ~(uint32_t)PRT_LIGHT_TABLE_BIT, p);
        const DLightNode& ln = lt.nodes[node];
        if ((double)p < ln.left_area) node = ln.left;
        else {
            p = (float)((double)p - ln.left_area);
            node = ln.right;
        }
    }
    return ~node;
}

static void build_light_tables(LightTree& out) {
    const std::vector<DLightNode>& full = out.full_nodes;
    const size_t nn = full.size();
    if (out.full_root < 0 || nn == 0) return;
    // per node: leaves of the subtree (contiguous in CDF order: the flattening is depth-first, left to right), its area as
    // the reference sums it, and whether TraverseSample is monotone on it (no span-1 node over a node inside)
    std::vector<uint32_t> first(nn, 0), count(nn, 0);
    std::vector<uint8_t> clean(nn, 1);
    std::vector<double> area(nn, 0.0);
    // children come after their parent in the breadth-first numbering: one backward sweep is a post-order
    auto leaf_first = [&](int32_t ref) { return ref < 0 ? (uint32_t)~ref : first[ref]; };
    auto leaf_count = [&](int32_t ref) { return ref < 0 ? 1u : count[ref]; };
    auto ref_area = [&](int32_t ref) { return ref < 0 ? out.tris[~ref].area : area[ref]; };
    for (size_t k = nn; k-- > 0;) {
        const DLightNode& n = full[k];
        const bool single = n.left == n.right;
        first[k] = leaf_first(n.left);
        count[k] = single ? leaf_count(n.left) : leaf_count(n.left) + leaf_count(n.right);
        area[k] = single ? ref_area(n.left) : ref_area(n.left) + ref_area(n.right); // BVH.cpp:21-45
        clean[k] = !(single && n.left >= 0) && (n.left < 0 || clean[n.left]) && (n.right < 0 || clean[n.right]);
    }
    std::vector<uint32_t> hdr, body; // headers first (8 words each), then thresholds / buckets; offsets fixed up at the end
    struct Made { int32_t node; uint32_t table; };
    std::vector<Made> made;
    std::vector<int32_t> stack{out.full_root};
    while (!stack.empty()) {
        const int32_t k = stack.back();
        stack.pop_back();
        if (clean[k] && count[k] >= PRT_LIGHT_TABLE_MIN) {
            const uint32_t n = count[k], f0 = first[k];
            DLightTable t{};
            t.first = f0;
            t.n = n;
            t.thr_off = (uint32_t)body.size();
            // thr[i]: the smallest p whose descent reaches leaf f0 + i or one to its right
            body.push_back(float_bits(0.f));
            for (uint32_t i = 1; i < n; ++i)
                body.push_back(first_pattern([&](float p) { return descend(full, k, p) >= (int32_t)(f0 + i); }));
            body.push_back(0x7f800000u); // thr[n] = +inf ends every walk
            uint32_t nb = 1;
            while (nb < 2 * n && nb < (1u << 20)) nb <<= 1;
            t.n_bkt = nb;
            t.inv_w = (float)((double)nb / area[k]);
            if (!(t.inv_w > 0.f) || !std::isfinite(t.inv_w)) { // degenerate areas: no table for this subtree
                body.resize(t.thr_off);
                if (full[k].left >= 0) stack.push_back(full[k].left);
                if (full[k].right >= 0 && full[k].right != full[k].left) stack.push_back(full[k].right);
                continue;
            }
            t.bkt_off = (uint32_t)body.size();
            const uint32_t* thr = nullptr;
            for (uint32_t b = 0; b < nb; ++b) {
                const uint32_t pat = first_pattern([&](float p) { return bucket_of(p, t.inv_w, nb) >= b; });
                uint32_t i = n - 1; // a bucket no p falls into: anything valid
                if (pat != 0x7f800000u) i = (uint32_t)descend(full, k, bits_float(pat)) - f0;
                thr = body.data() + t.thr_off;
                const uint32_t next = thr[i + 1];
                body.push_back(i);
                body.push_back(next);
            }
            made.push_back({k, (uint32_t)(hdr.size() / 8)});
            uint32_t w[8];
            std::memcpy(w, &t, sizeof(t));
            hdr.insert(hdr.end(), w, w + 8);
            continue;
        }
        if (full[k].left >= 0) stack.push_back(full[k].left);
        if (full[k].right >= 0 && full[k].right != full[k].left) stack.push_back(full[k].right);
    }
    if (made.empty()) return;
    const uint32_t shift = (uint32_t)hdr.size();
    for (size_t t = 0; t < made.size(); ++t) {
        hdr[8 * t + 2] += shift; // thr_off
        hdr[8 * t + 3] += shift; // bkt_off
    }
    std::vector<uint32_t> tab(hdr);
    tab.insert(tab.end(), body.begin(), body.end());
    // the part of the tree above the tables, renumbered breadth-first (K3 stages it in LDS: a handful of nodes)
    std::vector<int32_t> table_of(nn, -1);
    for (const Made& m : made) table_of[m.node] = (int32_t)m.table;
    auto map_ref = [&](int32_t ref, const std::vector<int32_t>& newidx) {
        if (ref < 0) return ref;
        if (table_of[ref] >= 0) return (int32_t)(PRT_LIGHT_TABLE_BIT | (uint32_t)table_of[ref]);
        return newidx[ref];
    };
    std::vector<int32_t> order, newidx(nn, -1);
    if (table_of[out.full_root] < 0) order.push_back(out.full_root);
    for (size_t q = 0; q < order.size(); ++q) {
        const DLightNode& n = full[order[q]];
        if (n.left >= 0 && table_of[n.left] < 0) order.push_back(n.left);
        if (n.right >= 0 && n.right != n.left && table_of[n.right] < 0) order.push_back(n.right);
    }
    for (size_t i = 0; i < order.size(); ++i) newidx[order[i]] = (int32_t)i;
    std::vector<DLightNode> top(order.size());
    for (size_t i = 0; i < order.size(); ++i) {
        DLightNode n = full[order[i]];
        n.left = map_ref(n.left, newidx);
        n.right = map_ref(n.right, newidx);
        top[i] = n;
    }
    LightTree cand = out;
    cand.nodes = top;
    cand.root = map_ref(out.full_root, newidx);
    cand.tab = tab;
    cand.n_tables = (uint32_t)made.size();
    // Verification: the table pick must be the descent's at every threshold and one pattern below it, at both ends of every
    // bucket, at 0, at the areas, far beyond them, and at 2^16 random patterns per table — through the WHOLE tree (the top
    // part included), against the full tree.  Any difference and the scene keeps the tree.
    auto same = [&](float p) { return light_pick(cand, p) == descend(full, out.full_root, p); };
    bool ok = true;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    const float total = (float)out.area;
    for (const Made& m : made) {
        DLightTable t;
        std::memcpy(&t, tab.data() + 8 * (size_t)m.table, sizeof(t));
        // p as the table sees it is what is left of the root's p after the top part's subtractions: test the table's own
        // domain through table_pick / descend on the subtree, and the whole pick through the root
        auto same_sub = [&](float p) { return table_pick(tab.data(), m.table, p) == descend(full, m.node, p); };
        for (uint32_t i = 0; i <= t.n && ok; ++i) {
            const uint32_t b = tab[t.thr_off + i];
            if (b != 0x7f800000u) ok = ok && same_sub(bits_float(b));
            if (b != 0 && b <= 0x7f800000u) ok = ok && same_sub(bits_float(b - 1));
            if (b < 0x7f7fffffu) ok = ok && same_sub(bits_float(b + 1));
        }
        for (uint32_t b = 0; b < t.n_bkt && ok; ++b) {
            const uint32_t pat = first_pattern([&](float p) { return bucket_of(p, t.inv_w, t.n_bkt) >= b; });
            if (pat == 0x7f800000u) continue;
            ok = ok && same_sub(bits_float(pat));
            if (pat) ok = ok && same_sub(bits_float(pat - 1));
        }
        const uint32_t top_pat = float_bits((float)(2.0 * area[m.node]));
        for (int r = 0; r < 65536 && ok; ++r) {
            rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
            ok = ok && same_sub(bits_float((uint32_t)(rng % ((uint64_t)top_pat + 1))));
        }
        ok = ok && same_sub(0.f) && same_sub((float)area[m.node]) && same_sub(3.0e38f) && same_sub(1e-30f);
    }
    const uint32_t tot_pat = float_bits(total);
    for (int r = 0; r < 262144 && ok; ++r) { // the whole pick, over the patterns p = (float)(sqrt(xi) * area) can take
        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
        ok = ok && same(bits_float((uint32_t)(rng % ((uint64_t)tot_pat + 1))));
    }
    for (uint32_t b = tot_pat > 64 ? tot_pat - 64 : 0; b <= tot_pat + 64 && ok; ++b) ok = ok && same(bits_float(b)); // the wrap-around region
    ok = ok && same(0.f);
    if (!ok) return;
    out.nodes.swap(cand.nodes);
    out.root = cand.root;
    out.tab.swap(cand.tab);
    out.n_tables = cand.n_tables;
}

void build_light_tree(const PrtSceneDesc& d, const std::vector<HostTri>& tris, const std::vector<DMaterial>& mats,
                      LightTree& out) {
    out.nodes.clear();
    out.tris.clear();
    out.root = -1;
    out.area = 0.0;
    RefBuilder rb(tris);
    std::vector<Obj> lights;
    // the lights list: what the caller handed to Camera::Render, or what main.cpp:40-45 builds (every emissive mesh)
    std::vector<uint32_t> list;
    if (d.light_meshes) list.assign(d.light_meshes, d.light_meshes + d.n_light_meshes);
    else
        for (uint32_t m = 0; m < d.n_meshes; ++m)
            if (mats[d.mesh_material[m]].has_emission) list.push_back(m);
    for (uint32_t m : list) {
        std::vector<Obj> objs;
        for (uint64_t t = d.mesh_first_tri[m]; t < d.mesh_first_tri[m + 1]; ++t) objs.push_back(Obj{(int32_t)t, -1});
        if (objs.empty()) continue;
        rb.build(objs, 0, objs.size());                               // world.Add(BVHNode(mesh))  main.cpp:39
        lights.push_back(Obj{-1, rb.build(objs, 0, objs.size())});    // lights.Add(BVHNode(mesh)) main.cpp:41
    }
    if (lights.empty()) return;
    const int32_t top = rb.build(lights, 0, lights.size()); // lights = HittableList(BVHNode(lights)) main.cpp:45
    out.area = rb.nodes[top].area;
    out.root = flatten_light(rb, Obj{-1, top}, out, tris);
    // Breadth-first numbering: the first K nodes are the top levels, which K3 stages in LDS (the descent is a chain
    // of dependent reads, one per level).  The leaf (CDF) order and every area stay as built.
    if (out.root >= 0) {
        std::vector<int32_t> order{out.root}, newidx(out.nodes.size(), -1);
        for (size_t q = 0; q < order.size(); ++q) {
            const DLightNode& n = out.nodes[order[q]];
            if (n.left >= 0) order.push_back(n.left);
            if (n.right >= 0 && n.right != n.left) order.push_back(n.right); // (a span-1 node lists its child twice)
        }
        for (size_t i = 0; i < order.size(); ++i) newidx[order[i]] = (int32_t)i;
        std::vector<DLightNode> renum(order.size());
        for (size_t i = 0; i < order.size(); ++i) {
            DLightNode n = out.nodes[order[i]];
            if (n.left >= 0) n.left = newidx[n.left];
            if (n.right >= 0) n.right = newidx[n.right];
            renum[i] = n;
        }
        out.nodes.swap(renum);
        out.root = 0;
    }
    out.full_nodes = out.nodes;
    out.full_root = out.root;
    out.tab.clear();
    out.n_tables = 0;
    build_light_tables(out);
    for (DLightTri& lt : out.tris) { // Triangle::Sample pdf = 1/area; TraverseSample pdf *= area; BVHNode::Sample pdf /= total
        double pdf = 1.0 / lt.area;
        pdf *= lt.area;
        pdf /= out.area;
        lt.pdf = pdf;
    }
}

void setup_camera(const PrtCamera& c, DCamera& out) { // Camera.cpp:75-106
    const int W = c.width < 1 ? 1 : c.width, H = c.height < 1 ? 1 : c.height;
    const double aspect = double(W) / double(H);
    V eye = load(c.eye), look = load(c.look_at), up = load(c.up);
    V el = sub(eye, look);
    const double focal = std::sqrt(dot(el, el));
    const double theta = c.fovy * 0.01745329251994329576923690768489; // glm::radians
    const double h = std::tan(theta / 2.0);
    const double vh = 2. * h * focal;
    const double vw = vh * aspect;
    V w = unit(el);
    V u = unit(cross(up, w));
    V v = cross(w, u);
    V vu{vw * u.x, vw * u.y, vw * u.z};
    V nv{-v.x, -v.y, -v.z};
    V vv{vh * nv.x, vh * nv.y, vh * nv.z};
    V du{vu.x / (double)W, vu.y / (double)W, vu.z / (double)W};
    V dv{vv.x / (double)H, vv.y / (double)H, vv.z / (double)H};
    V fw{focal * w.x, focal * w.y, focal * w.z};
    V ul = sub(sub(sub(eye, fw), V{vu.x / 2., vu.y / 2., vu.z / 2.}), V{vv.x / 2., vv.y / 2., vv.z / 2.});
    V s = add(du, dv);
    V p00 = add(ul, V{0.5 * s.x, 0.5 * s.y, 0.5 * s.z});
    store(out.center, eye);
    store(out.pixel00, p00);
    store(out.du, du);
    store(out.dv, dv);
    out.width = W;
    out.height = H;
}

} // namespace prt

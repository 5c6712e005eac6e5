// Developer tool (CPU only): would an 8-wide node whose traversal order comes from the RAY OCTANT (no per-visit sort, one
// stack entry per node: node + hit mask — the compressed-wide-BVH scheme) make fewer node visits per ray than the product's
// sorted 4-wide node?  VERDICT r3 #5: "build it in both builders only if the simulation shows >= 30 % fewer visits".
//
// Everything is derived from the product's own binary SAH tree (BuiltBVH::keep_binary), so the leaves and the triangle order
// are the product's.  Formats:
//   4-sorted   the product: 4-wide collapse, 16-bit boxes, children entered near to far (exact sort), one stack entry per child
//   8-sorted   8-wide collapse (same greedy largest-box rule), 8-bit boxes in a per-node frame, exact sort — the upper bound
//              on what ANY ordering of the 8-wide node can reach
//   8-octant   same nodes; child slots assigned at build time so that slot ^ octant is the visiting priority (greedy
//              assignment on (child centre - node centre) . octant direction, as in Ylitie et al. 2017); traversal
//              visits the hit children in that fixed order; ONE stack entry per node (ref + mask of hit children not yet
//              entered)
// Output per format: node visits, leaf visits, triangle tests, distinct 128-byte lines missed in a 4 MB LRU L2 model, box tests
// and maximum / mean stack entries per ray.
//   g++ -O2 -std=c++17 -fopenmp tools/sim_oct8.cpp pooraytracer_amd/csrc/bvh_build.cpp pooraytracer_amd/csrc/scene_setup.cpp -o /tmp/sim_oct8
//   /tmp/sim_oct8 soup 8000000 200000 | /tmp/sim_oct8 file tris.f64 200000      (tris.f64: [n][3][3] doubles, tools/export_tris.py)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>

#include "../pooraytracer_amd/csrc/prt_host.h"

using namespace prt;

struct WBox {
    float lo[3], hi[3];
};
struct WNode {
    int nk = 0;
    WBox box[8];
    int32_t ref[8];   // >= 0: node index of this variant, < 0: the product's leaf ref; 0x80000000: empty slot (octant layout)
    bool used[8] = {false, false, false, false, false, false, false, false};
};

struct Cache { // set-associative LRU over 128-byte lines: one XCD's 4 MB L2
    size_t sets, ways;
    std::vector<uint64_t> tag;
    std::vector<uint32_t> age;
    uint32_t clock = 0;
    uint64_t misses = 0;
    Cache(size_t bytes, size_t ways_) : sets(bytes / 128 / ways_), ways(ways_), tag(sets * ways_, ~0ULL), age(sets * ways_, 0) {}
    void access(uint64_t line) {
        const size_t s = (size_t)((line * 0x9E3779B97F4A7C15ULL) >> 20) % sets;
        uint64_t* t = &tag[s * ways];
        uint32_t* a = &age[s * ways];
        ++clock;
        size_t victim = 0;
        for (size_t w = 0; w < ways; ++w) {
            if (t[w] == line) {
                a[w] = clock;
                return;
            }
            if (a[w] < a[victim]) victim = w;
        }
        t[victim] = line;
        a[victim] = clock;
        ++misses;
    }
};

struct Ray {
    double o[3], d[3];
};

static bool tri_hit(const HostTri& T, const Ray& r, double tmin, double tmax, double& t_out) {
    double e1[3], e2[3], p[3], s[3], q[3];
    for (int a = 0; a < 3; ++a) {
        e1[a] = T.v[1][a] - T.v[0][a];
        e2[a] = T.v[2][a] - T.v[0][a];
    }
    p[0] = r.d[1] * e2[2] - r.d[2] * e2[1];
    p[1] = r.d[2] * e2[0] - r.d[0] * e2[2];
    p[2] = r.d[0] * e2[1] - r.d[1] * e2[0];
    const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (std::fabs(det) < 1e-300) return false;
    const double inv = 1.0 / det;
    for (int a = 0; a < 3; ++a) s[a] = r.o[a] - T.v[0][a];
    const double u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * inv;
    if (u < 0 || u > 1) return false;
    q[0] = s[1] * e1[2] - s[2] * e1[1];
    q[1] = s[2] * e1[0] - s[0] * e1[2];
    q[2] = s[0] * e1[1] - s[1] * e1[0];
    const double v = (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]) * inv;
    if (v < 0 || u + v > 1) return false;
    const double t = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    if (t <= tmin || t >= tmax) return false;
    t_out = t;
    return true;
}

static double area(const WBox& b) {
    const double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
    return x * y + y * z + z * x;
}

// Collapse of the binary tree into `width`-wide nodes: a node absorbs its inner child with the largest box until it is full
// (the product's rule, bvh_build.cpp; no stack budget here — the octant scheme needs one entry per NODE, the sorted schemes
// report what they would need).  Inner children get consecutive indices, depth-first.
// `budget` > 0: the product's stack rule for schemes that push one entry per hit child — a node with k children leaves
// budget - (k-1) entries to each child's subtree, and a child is absorbed only while every resulting child's binary subtree is
// no taller than what is left (bvh_build.cpp).
static std::vector<WNode> collapse(const std::vector<BuiltBVH::BinNode>& bin, int width, int budget = 0) {
    std::vector<uint8_t> h2(bin.size(), 1);
    for (size_t i = bin.size(); i-- > 0;) {
        const int a = bin[i].ref[0] >= 0 ? h2[bin[i].ref[0]] : 0, c = bin[i].ref[1] >= 0 ? h2[bin[i].ref[1]] : 0;
        h2[i] = (uint8_t)(1 + std::max(a, c));
    }
    auto height = [&](int32_t ref) { return ref >= 0 ? (int)h2[ref] : 0; };
    std::vector<WNode> out;
    struct Open { int32_t bin; int32_t slot; int budget; };
    std::vector<Open> todo;
    out.emplace_back();
    todo.push_back({0, 0, budget});
    while (!todo.empty()) {
        const Open o = todo.back();
        todo.pop_back();
        WNode n;
        int32_t bref[8];
        auto put = [&](int k, const BuiltBVH::BinNode& b, int side) {
            for (int a = 0; a < 3; ++a) {
                n.box[k].lo[a] = b.lo[side][a];
                n.box[k].hi[a] = b.hi[side][a];
            }
            bref[k] = b.ref[side];
        };
        put(0, bin[o.bin], 0);
        put(1, bin[o.bin], 1);
        n.nk = 2;
        while (n.nk < width) {
            int best = -1;
            double ba = -1;
            for (int k = 0; k < n.nk; ++k) {
                if (bref[k] < 0 || !(area(n.box[k]) > ba)) continue;
                if (budget > 0) {
                    const int left = o.budget - n.nk;
                    bool fits = height(bin[bref[k]].ref[0]) <= left && height(bin[bref[k]].ref[1]) <= left;
                    for (int j = 0; j < n.nk && fits; ++j)
                        if (j != k && height(bref[j]) > left) fits = false;
                    if (!fits) continue;
                }
                ba = area(n.box[k]);
                best = k;
            }
            if (best < 0) break;
            const BuiltBVH::BinNode& c = bin[bref[best]];
            put(best, c, 0);
            put(n.nk, c, 1);
            n.nk++;
        }
        for (int k = 0; k < n.nk; ++k) {
            n.used[k] = true;
            if (bref[k] < 0) n.ref[k] = bref[k];
            else {
                n.ref[k] = (int32_t)out.size();
                out.emplace_back();
            }
        }
        for (int k = n.nk - 1; k >= 0; --k)
            if (bref[k] >= 0) todo.push_back({bref[k], n.ref[k], o.budget - (n.nk - 1)});
        out[o.slot] = n;
    }
    return out;
}


// Cost-optimal collapse (Ylitie, Karras, Laine 2017, §3): minimise the summed surface area of the WIDE nodes — a visit costs a
// SIMT kernel the same whatever the node holds, and the area is the probability of the visit.  c[n][i] = least cost of turning
// the binary subtree n into a forest of at most i wide-tree roots (i = 1: a single wide node, or the binary leaf itself);
// the children of a wide node rooted at n are a forest of `width` roots of n's two subtrees.  Leaves are the builder's.
static std::vector<WNode> collapse_optimal(const std::vector<BuiltBVH::BinNode>& bin, int width) {
    const size_t nb = bin.size();
    auto box_area = [&](size_t i) { // area of binary node i = union of its two child boxes
        WBox b;
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = std::min(bin[i].lo[0][a], bin[i].lo[1][a]);
            b.hi[a] = std::max(bin[i].hi[0][a], bin[i].hi[1][a]);
        }
        return area(b);
    };
    const int W = width;
    std::vector<double> c(nb * (W + 1), 0.0);      // c[n * (W+1) + i], i = 1..W
    std::vector<int8_t> split(nb * (W + 1), 0);    // forest of i roots: how many go to the left subtree (0 = "use i - 1")
    auto cost = [&](int32_t ref, int i) -> double { return ref < 0 ? 0.0 : c[(size_t)ref * (W + 1) + std::min(i, W)]; };
    for (size_t n = nb; n-- > 0;) { // children follow their parent in the pre-order array
        const int32_t l = bin[n].ref[0], r = bin[n].ref[1];
        auto distribute = [&](int j, int8_t& best_a) { // forest of j >= 2 roots over the two subtrees
            double best = 1e300;
            for (int a = 1; a < j; ++a) {
                const double v = cost(l, a) + cost(r, j - a);
                if (v < best) {
                    best = v;
                    best_a = (int8_t)a;
                }
            }
            return best;
        };
        int8_t a1 = 1;
        c[n * (W + 1) + 1] = box_area(n) + distribute(W, a1); // one wide node with up to W children
        split[n * (W + 1) + 1] = a1;
        for (int i = 2; i <= W; ++i) {
            int8_t ai = 1;
            const double d = distribute(i, ai);
            if (d < c[n * (W + 1) + i - 1]) {
                c[n * (W + 1) + i] = d;
                split[n * (W + 1) + i] = ai;
            } else {
                c[n * (W + 1) + i] = c[n * (W + 1) + i - 1];
                split[n * (W + 1) + i] = 0;
            }
        }
    }
    // emit: children of the wide node at binary node n = forest(n, W)
    std::vector<WNode> out;
    struct Kid { int32_t ref; WBox box; };
    std::vector<Kid> kids;
    // forest(ref with box, i): appends the roots
    std::function<void(int32_t, const WBox&, int)> forest = [&](int32_t ref, const WBox& box, int i) {
        if (ref < 0) {
            kids.push_back({ref, box});
            return;
        }
        i = std::min(i, W);
        while (i > 1 && split[(size_t)ref * (W + 1) + i] == 0) --i;
        if (i == 1) {
            kids.push_back({ref, box}); // a wide node of its own
            return;
        }
        const int a = split[(size_t)ref * (W + 1) + i];
        WBox bl, br;
        for (int x = 0; x < 3; ++x) {
            bl.lo[x] = bin[ref].lo[0][x]; bl.hi[x] = bin[ref].hi[0][x];
            br.lo[x] = bin[ref].lo[1][x]; br.hi[x] = bin[ref].hi[1][x];
        }
        forest(bin[ref].ref[0], bl, a);
        forest(bin[ref].ref[1], br, i - a);
    };
    struct Open { int32_t bin; int32_t slot; };
    std::vector<Open> todo;
    out.emplace_back();
    todo.push_back({0, 0});
    while (!todo.empty()) {
        const Open o = todo.back();
        todo.pop_back();
        kids.clear();
        const int a = split[(size_t)o.bin * (W + 1) + 1];
        WBox bl, br;
        for (int x = 0; x < 3; ++x) {
            bl.lo[x] = bin[o.bin].lo[0][x]; bl.hi[x] = bin[o.bin].hi[0][x];
            br.lo[x] = bin[o.bin].lo[1][x]; br.hi[x] = bin[o.bin].hi[1][x];
        }
        forest(bin[o.bin].ref[0], bl, a);
        forest(bin[o.bin].ref[1], br, W - a);
        WNode n;
        n.nk = (int)kids.size();
        std::vector<Kid> mine = kids; // (forest() reuses `kids`)
        for (int k = 0; k < n.nk; ++k) {
            n.used[k] = true;
            n.box[k] = mine[k].box;
            if (mine[k].ref < 0) n.ref[k] = mine[k].ref;
            else {
                n.ref[k] = (int32_t)out.size();
                out.emplace_back();
            }
        }
        for (int k = n.nk - 1; k >= 0; --k)
            if (mine[k].ref >= 0) todo.push_back({mine[k].ref, n.ref[k]});
        out[o.slot] = n;
    }
    return out;
}

// child boxes on a per-node grid of `bits` bits per coordinate (origin = the node's lower corner snapped to the global 16-bit
// grid, step = a power of two of the global step), rounded outward
static void quantise(std::vector<WNode>& nodes, const float g0[3], const float gs[3], int bits) {
    const double top = (double)((1 << bits) - 1);
    for (WNode& n : nodes)
        for (int a = 0; a < 3; ++a) {
            double lo = 1e300, hi = -1e300;
            for (int k = 0; k < 8; ++k)
                if (n.used[k]) {
                    lo = std::min(lo, (double)n.box[k].lo[a]);
                    hi = std::max(hi, (double)n.box[k].hi[a]);
                }
            const double qlo = std::floor((lo - g0[a]) / gs[a]), qhi = std::ceil((hi - g0[a]) / gs[a]);
            int e = 0;
            while ((qhi - qlo) > top * std::ldexp(1.0, e)) ++e;
            const double step = std::ldexp(1.0, e) * gs[a], org = g0[a] + qlo * gs[a];
            for (int k = 0; k < 8; ++k)
                if (n.used[k]) {
                    const double l8 = std::floor((n.box[k].lo[a] - org) / step), h8 = std::ceil((n.box[k].hi[a] - org) / step);
                    n.box[k].lo[a] = std::nextafter((float)(org + std::max(0.0, l8) * step), -1e30f);
                    n.box[k].hi[a] = std::nextafter((float)(org + std::min(top, std::max(h8, l8 + 1)) * step), 1e30f);
                }
        }
}

// Octant slots: child c goes to slot s such that the sum over children of (centre_c - centre_node) . dir(s) is large, greedily
// (largest remaining score first).  dir(s) = (s&1 ? -1 : +1, s&2 ? -1 : +1, s&4 ? -1 : +1); a ray with octant q (bit a set when
// d[a] < 0) enters the hit children in DESCENDING order of slot ^ q (near corner first).
static void assign_octant_slots(std::vector<WNode>& nodes) {
    for (WNode& n : nodes) {
        double c0[3] = {0, 0, 0};
        WBox all;
        for (int a = 0; a < 3; ++a) {
            all.lo[a] = 1e30f;
            all.hi[a] = -1e30f;
        }
        for (int k = 0; k < n.nk; ++k)
            for (int a = 0; a < 3; ++a) {
                all.lo[a] = std::min(all.lo[a], n.box[k].lo[a]);
                all.hi[a] = std::max(all.hi[a], n.box[k].hi[a]);
            }
        for (int a = 0; a < 3; ++a) c0[a] = 0.5 * ((double)all.lo[a] + all.hi[a]);
        double score[8][8];
        for (int k = 0; k < n.nk; ++k)
            for (int s = 0; s < 8; ++s) {
                double v = 0;
                for (int a = 0; a < 3; ++a) {
                    const double c = 0.5 * ((double)n.box[k].lo[a] + n.box[k].hi[a]) - c0[a];
                    v += ((s >> a) & 1) ? -c : c;
                }
                score[k][s] = v;
            }
        int slot_of[8];
        bool child_done[8] = {false, false, false, false, false, false, false, false}, slot_taken[8] = {false, false, false, false, false, false, false, false};
        for (int it = 0; it < n.nk; ++it) {
            int bk = -1, bs = -1;
            double bv = -1e300;
            for (int k = 0; k < n.nk; ++k)
                if (!child_done[k])
                    for (int s = 0; s < 8; ++s)
                        if (!slot_taken[s] && score[k][s] > bv) {
                            bv = score[k][s];
                            bk = k;
                            bs = s;
                        }
            child_done[bk] = true;
            slot_taken[bs] = true;
            slot_of[bk] = bs;
        }
        WNode m;
        m.nk = n.nk;
        for (int s = 0; s < 8; ++s) {
            m.used[s] = false;
            m.ref[s] = (int32_t)0x80000000;
        }
        for (int k = 0; k < n.nk; ++k) {
            m.box[slot_of[k]] = n.box[k];
            m.ref[slot_of[k]] = n.ref[k];
            m.used[slot_of[k]] = true;
        }
        n = m;
    }
}

struct Stats {
    double visits = 0, leaves = 0, tris = 0, lines = 0, boxes = 0, stack_mean = 0;
    int max_stack = 0;
};

enum Order { SORTED, OCTANT, OCT_CHILD, OCT_GROUP, SORTED_CULL };
// SORTED_CULL: the product's scheme with the child's entry distance kept beside its ref: a popped child whose entry distance lies
// beyond the closest hit found meanwhile is dropped without a visit.
// OCT_CHILD: octant order, one stack entry per hit child (the product's stack, no sort), no re-test at pop.
// OCT_GROUP: a node's hit LEAVES are entered first (one entry each), then its hit inner children from ONE group entry in octant
// order; no re-test at pop (the entry carries no box).

static Stats run(const std::vector<WNode>& nodes, int node_bytes, Order order, const BuiltBVH& B, const std::vector<HostTri>& tris,
                 const std::vector<Ray>& rays, size_t warm) {
    Cache L2(4u << 20, 16);
    Stats st;
    const uint64_t tri_base = 1ULL << 40;
    std::vector<std::pair<float, int32_t>> hitk;
    struct Entry { int32_t ref; uint32_t mask; float tn = 0.f; }; // sorted scheme: one child per entry (mask unused); octant scheme: node + remaining hit slots
    std::vector<Entry> stack;
    double stack_sum = 0, stack_samples = 0;
    for (size_t ri = 0; ri < rays.size(); ++ri) {
        if (ri == warm) {
            L2.misses = 0;
            st = Stats();
            stack_sum = stack_samples = 0;
        }
        const Ray& r = rays[ri];
        double inv[3];
        for (int a = 0; a < 3; ++a) inv[a] = 1.0 / r.d[a];
        const int oct = (r.d[0] < 0 ? 1 : 0) | (r.d[1] < 0 ? 2 : 0) | (r.d[2] < 0 ? 4 : 0);
        double tbest = 1e300;
        stack.clear();
        int32_t cur = 0;
        auto leaf = [&](int32_t ref) {
            const uint32_t enc = ~(uint32_t)ref, first = enc >> 3, cnt = (enc & 7u) + 1u;
            st.leaves++;
            for (uint32_t i = first; i < first + cnt; ++i) {
                st.tris++;
                L2.access(tri_base + i);
                double t;
                if (tri_hit(tris[B.order[i]], r, 1e-4, tbest, t)) tbest = t;
            }
        };
        for (;;) {
            if (cur >= 0) {
                const WNode& n = nodes[cur];
                st.visits++;
                const uint64_t a0 = (uint64_t)cur * node_bytes, a1 = a0 + node_bytes - 1;
                for (uint64_t l = a0 / 128; l <= a1 / 128; ++l) L2.access(l);
                hitk.clear();
                uint32_t mask = 0;
                for (int i = 0; i < 8; ++i) {
                    if (!n.used[i]) continue;
                    st.boxes++;
                    double tn = 1e-4, tf = tbest;
                    for (int a = 0; a < 3; ++a) {
                        double t0 = ((double)n.box[i].lo[a] - r.o[a]) * inv[a], t1 = ((double)n.box[i].hi[a] - r.o[a]) * inv[a];
                        if (t0 > t1) std::swap(t0, t1);
                        tn = std::max(tn, t0);
                        tf = std::min(tf, t1);
                    }
                    if (tn <= tf) {
                        hitk.push_back({(float)tn, n.ref[i]});
                        mask |= 1u << i;
                    }
                }
                if (order == SORTED || order == SORTED_CULL) {
                    std::sort(hitk.begin(), hitk.end(), [](auto& x, auto& y) { return x.first < y.first; });
                    for (size_t i = hitk.size(); i-- > 1;) stack.push_back({hitk[i].second, 0, hitk[i].first});
                    cur = hitk.empty() ? (int32_t)0x80000000 : hitk[0].second;
                } else if (order == OCT_CHILD) {
                    // far to near by priority (slot ^ oct ascending = far first), nearest entered directly
                    int32_t first = (int32_t)0x80000000;
                    for (int p = 0; p < 8; ++p) {
                        const int i = p ^ oct;
                        if ((mask >> i) & 1u) {
                            if (first != (int32_t)0x80000000) stack.push_back({first, 0});
                            first = n.ref[i];
                        }
                    }
                    cur = first; // the last one found = highest priority = nearest corner
                } else if (order == OCT_GROUP) {
                    uint32_t inner = 0, leafm = 0;
                    for (int i = 0; i < 8; ++i)
                        if ((mask >> i) & 1u) (n.ref[i] >= 0 ? inner : leafm) |= 1u << i;
                    if (inner) stack.push_back({cur, inner | 0x100u}); // group entry (flag bit 8)
                    for (int p = 0; p < 8; ++p) { // leaves on top, nearest popped first
                        const int i = p ^ oct;
                        if ((leafm >> i) & 1u) stack.push_back({n.ref[i], 0});
                    }
                    cur = (int32_t)0x80000000;
                } else {
                    // the node's hit children are entered in descending order of slot ^ oct (see the pop below)
                    if (mask) stack.push_back({cur, mask});
                    cur = (int32_t)0x80000000;
                }
            } else if (cur != (int32_t)0x80000000) {
                leaf(cur);
                cur = (int32_t)0x80000000;
            }
            st.max_stack = std::max(st.max_stack, (int)stack.size());
            stack_sum += (double)stack.size();
            stack_samples += 1;
            if (cur == (int32_t)0x80000000) {
                if (stack.empty()) break;
                if (order == SORTED_CULL) {
                    cur = (int32_t)0x80000000;
                    while (!stack.empty() && cur == (int32_t)0x80000000) {
                        if ((double)stack.back().tn <= tbest) cur = stack.back().ref;
                        stack.pop_back();
                    }
                    if (cur == (int32_t)0x80000000) break;
                } else if (order == SORTED || order == OCT_CHILD || (order == OCT_GROUP && !(stack.back().mask & 0x100u))) {
                    cur = stack.back().ref;
                    stack.pop_back();
                } else if (order == OCT_GROUP) {
                    Entry& e = stack.back();
                    const WNode& n = nodes[e.ref];
                    int best = -1, bp = -1;
                    for (int i = 0; i < 8; ++i)
                        if ((e.mask >> i) & 1u) {
                            const int p = i ^ oct;
                            if (p > bp) {
                                bp = p;
                                best = i;
                            }
                        }
                    e.mask &= ~(1u << best);
                    cur = n.ref[best];
                    if (!(e.mask & 0xffu)) stack.pop_back();
                } else {
                    Entry& e = stack.back();
                    const WNode& n = nodes[e.ref];
                    // next child of that node: LARGEST slot ^ oct among the remaining hits — a slot's bit a is set when the child
                    // lies towards -axis a, the octant's when the ray runs towards -axis a: where they differ the child is on the
                    // side the ray comes from, so slot ^ oct = 7 is the nearest corner and 0 the farthest
                    int best = -1, bp = -1;
                    for (int i = 0; i < 8; ++i)
                        if ((e.mask >> i) & 1u) {
                            const int p = i ^ oct;
                            if (p > bp) {
                                bp = p;
                                best = i;
                            }
                        }
                    e.mask &= ~(1u << best);
                    cur = n.ref[best];
                    if (!e.mask) stack.pop_back();
                    // the child's box was tested against the tbest of the node visit: re-test against the current one (one box
                    // test, no fetch — the entry would carry nothing but the slot; the kernel would re-fetch or skip this)
                    double tn = 1e-4, tf = tbest;
                    for (int a = 0; a < 3; ++a) {
                        double t0 = ((double)n.box[best].lo[a] - r.o[a]) * inv[a], t1 = ((double)n.box[best].hi[a] - r.o[a]) * inv[a];
                        if (t0 > t1) std::swap(t0, t1);
                        tn = std::max(tn, t0);
                        tf = std::min(tf, t1);
                    }
                    if (!(tn <= tf)) cur = (int32_t)0x80000000; // culled by a hit found meanwhile
                }
            }
        }
    }
    const double n = (double)(rays.size() - warm);
    st.visits /= n;
    st.leaves /= n;
    st.tris /= n;
    st.boxes /= n;
    st.lines = (double)L2.misses / n;
    st.stack_mean = stack_sum / std::max(1.0, stack_samples);
    return st;
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "soup";
    std::mt19937_64 rng(4);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<HostTri> tris;
    size_t n_rays = 200000;
    if (mode == "soup") {
        const size_t n_tris = argc > 2 ? (size_t)std::atoll(argv[2]) : 8000000;
        if (argc > 3) n_rays = (size_t)std::atoll(argv[3]);
        tris.resize(n_tris);
        for (size_t i = 0; i < n_tris; ++i) {
            HostTri& T = tris[i];
            const double c[3] = {U(rng), U(rng), U(rng)};
            for (int v = 0; v < 3; ++v)
                for (int a = 0; a < 3; ++a) T.v[v][a] = c[a] + (U(rng) - 0.5) * 0.01;
        }
    } else {
        FILE* f = std::fopen(argv[2], "rb");
        if (!f) return 1;
        std::fseek(f, 0, SEEK_END);
        const size_t n_tris = (size_t)std::ftell(f) / 72;
        std::fseek(f, 0, SEEK_SET);
        tris.resize(n_tris);
        for (size_t i = 0; i < n_tris; ++i)
            if (std::fread(tris[i].v, 8, 9, f) != 9) return 1;
        std::fclose(f);
        if (argc > 3) n_rays = (size_t)std::atoll(argv[3]);
    }
    double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
    for (size_t i = 0; i < tris.size(); ++i) {
        HostTri& T = tris[i];
        for (int a = 0; a < 3; ++a) {
            T.lo[a] = std::min(T.v[0][a], std::min(T.v[1][a], T.v[2][a]));
            T.hi[a] = std::max(T.v[0][a], std::max(T.v[1][a], T.v[2][a]));
            if (T.hi[a] - T.lo[a] < 1e-4) { // Triangle.cpp: boxes padded to 1e-4
                T.lo[a] -= 5e-5;
                T.hi[a] += 5e-5;
            }
            blo[a] = std::min(blo[a], T.lo[a]);
            bhi[a] = std::max(bhi[a], T.hi[a]);
        }
        T.material = 0;
        T.prim = (int32_t)i;
    }
    BuiltBVH B;
    B.keep_binary = true;
    std::string err;
    if (!build_bvh(tris, B, &err)) {
        std::fprintf(stderr, "build failed: %s\n", err.c_str());
        return 1;
    }
    std::printf("%s: %zu triangles, binary tree %zu nodes, product 4-wide %zu nodes (stack need %d)\n", mode == "soup" ? "soup" : argv[2], tris.size(),
                B.binary.size(), B.nodes.size(), B.stack_need);
    std::vector<Ray> rays(n_rays);
    for (Ray& r : rays) { // origin uniform in the scene's box, direction uniform on the sphere
        for (int a = 0; a < 3; ++a) r.o[a] = blo[a] + U(rng) * (bhi[a] - blo[a]);
        const double z = 2 * U(rng) - 1, phi = 6.283185307179586 * U(rng), s = std::sqrt(std::max(0.0, 1 - z * z));
        r.d[0] = s * std::cos(phi);
        r.d[1] = s * std::sin(phi);
        r.d[2] = z;
    }
    struct Variant { std::string name; std::vector<WNode> nodes; int bytes; Order order; };
    std::vector<Variant> vs;
    {
        Variant v{"4-wide sorted, 64 B, 16-bit boxes (product)", collapse(B.binary, 4), 64, SORTED};
        quantise(v.nodes, B.grid_origin, B.grid_step, 16);
        vs.push_back(v);
    }
    {
        std::vector<WNode> n8 = collapse(B.binary, 8);
        Variant a{"8-wide sorted, 128 B, 16-bit boxes", n8, 128, SORTED};
        quantise(a.nodes, B.grid_origin, B.grid_step, 16);
        vs.push_back(a);
        Variant b{"8-wide sorted, 96 B, 8-bit boxes", n8, 96, SORTED};
        quantise(b.nodes, B.grid_origin, B.grid_step, 8);
        vs.push_back(b);
        Variant c{"8-wide OCTANT order, 96 B, 8-bit boxes, 1 entry/node", n8, 96, OCTANT};
        quantise(c.nodes, B.grid_origin, B.grid_step, 8);
        assign_octant_slots(c.nodes);
        vs.push_back(c);
        Variant d{"8-wide OCTANT order, 128 B, 16-bit boxes, 1 entry/node", n8, 128, OCTANT};
        quantise(d.nodes, B.grid_origin, B.grid_step, 16);
        assign_octant_slots(d.nodes);
        vs.push_back(d);
        Variant e{"8-wide octant, 128 B, leaves first + 1 GROUP entry/node, no re-test", d.nodes, 128, OCT_GROUP};
        vs.push_back(e);
        Variant f{"8-wide octant, 128 B, entry per CHILD, no re-test (unbudgeted)", d.nodes, 128, OCT_CHILD};
        vs.push_back(f);
        Variant g{"8-wide octant, 128 B, entry per CHILD, stack budget 40", collapse(B.binary, 8, 40), 128, OCT_CHILD};
        quantise(g.nodes, B.grid_origin, B.grid_step, 16);
        assign_octant_slots(g.nodes);
        vs.push_back(g);
        Variant h{"4-wide sorted, stack budget 40 (the product as shipped)", collapse(B.binary, 4, 40), 64, SORTED};
        quantise(h.nodes, B.grid_origin, B.grid_step, 16);
        vs.push_back(h);
        Variant hc{"4-wide sorted, stack budget 40, entry distance kept: CULL AT POP", h.nodes, 64, SORTED_CULL};
        vs.push_back(hc);
        Variant ho{"4-wide sorted, COST-OPTIMAL collapse (min summed node area)", collapse_optimal(B.binary, 4), 64, SORTED};
        quantise(ho.nodes, B.grid_origin, B.grid_step, 16);
        vs.push_back(ho);
        Variant i6{"6-wide octant, entry per CHILD, stack budget 40", collapse(B.binary, 6, 40), 96, OCT_CHILD};
        quantise(i6.nodes, B.grid_origin, B.grid_step, 16);
        assign_octant_slots(i6.nodes);
        vs.push_back(i6);
    }
    std::vector<Stats> res(vs.size());
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < vs.size(); ++i) res[i] = run(vs[i].nodes, vs[i].bytes, vs[i].order, B, tris, rays, n_rays / 3);
    for (size_t i = 0; i < vs.size(); ++i) {
        double kids = 0;
        for (const WNode& n : vs[i].nodes) kids += n.nk;
        std::printf("%-70s nodes %8zu (%.2f kids)  visits/ray %6.2f (%+5.1f %%)  box tests %6.1f  leaves %5.2f  tris %5.2f  L2-miss lines %5.1f  stack max %2d mean %.1f\n",
                    vs[i].name.c_str(), vs[i].nodes.size(), kids / vs[i].nodes.size(), res[i].visits, 100.0 * (res[i].visits / res[0].visits - 1.0), res[i].boxes,
                    res[i].leaves, res[i].tris, res[i].lines, res[i].max_stack, res[i].stack_mean);
    }
    return 0;
}

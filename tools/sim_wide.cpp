// Developer tool (CPU only): what would other node formats cost an S4-class scene in node visits and in L2-missing 128-byte
// lines per ray?  Builds the 8M-triangle soup's tree with the product's host builder (4-wide, 16-bit boxes), derives wider
// / compressed variants from it, traverses the same random rays through each (closest hit, children near to far, one
// stack entry per pushed child — the kernels' scheme) and runs every node / triangle fetch through an LRU model of one
// XCD's 4 MB L2.
//   g++ -O2 -std=c++17 -fopenmp tools/sim_wide.cpp pooraytracer_amd/csrc/bvh_build.cpp pooraytracer_amd/csrc/scene_setup.cpp -o /tmp/sim_wide
//   /tmp/sim_wide [n_tris=8000000] [n_rays=300000]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
    WBox box[16];
    int32_t ref[16]; // >= 0: index into the variant's node array, < 0: the product's leaf ref
};

struct Cache { // set-associative LRU over 128-byte lines
    size_t sets, ways;
    std::vector<uint64_t> tag;
    std::vector<uint32_t> age;
    uint32_t clock = 0;
    uint64_t hits = 0, misses = 0;
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
                ++hits;
                return;
            }
            if (a[w] < a[victim]) victim = w;
        }
        t[victim] = line;
        a[victim] = clock;
        ++misses;
    }
    void reset_counts() { hits = misses = 0; }
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

struct Variant {
    const char* name;
    std::vector<WNode> nodes;
    int node_bytes;     // bytes per node
    int align_children; // children of a node are consecutive: numbering done by number()
};

struct Stats {
    double visits = 0, tris = 0, leaves = 0, lines = 0, node_lines = 0;
    int max_stack = 0;
};

static Stats run(const Variant& V, const BuiltBVH& B, const std::vector<HostTri>& tris, const std::vector<Ray>& rays, size_t warm) {
    Cache L2(4u << 20, 16);
    Stats st;
    const uint64_t tri_base = 1ULL << 40;
    uint64_t node_misses = 0;
    std::vector<std::pair<float, int32_t>> hitk;
    std::vector<int32_t> stack;
    for (size_t ri = 0; ri < rays.size(); ++ri) {
        if (ri == warm) {
            L2.reset_counts();
            node_misses = 0;
            st = Stats();
        }
        const Ray& r = rays[ri];
        double inv[3];
        for (int a = 0; a < 3; ++a) inv[a] = 1.0 / r.d[a];
        double tbest = 1e300;
        stack.clear();
        int32_t cur = 0;
        for (;;) {
            if (cur >= 0) {
                const WNode& n = V.nodes[cur];
                st.visits++;
                {
                    const uint64_t a0 = (uint64_t)cur * V.node_bytes, a1 = a0 + V.node_bytes - 1;
                    for (uint64_t l = a0 / 128; l <= a1 / 128; ++l) {
                        const uint64_t m0 = L2.misses;
                        L2.access(l);
                        node_misses += L2.misses - m0;
                    }
                }
                hitk.clear();
                for (int i = 0; i < n.nk; ++i) {
                    double tn = 1e-4, tf = tbest;
                    for (int a = 0; a < 3; ++a) {
                        double t0 = ((double)n.box[i].lo[a] - r.o[a]) * inv[a], t1 = ((double)n.box[i].hi[a] - r.o[a]) * inv[a];
                        if (t0 > t1) std::swap(t0, t1);
                        tn = std::max(tn, t0);
                        tf = std::min(tf, t1);
                    }
                    if (tn <= tf) hitk.push_back({(float)tn, n.ref[i]});
                }
                std::sort(hitk.begin(), hitk.end(), [](auto& x, auto& y) { return x.first < y.first; });
                for (size_t i = hitk.size(); i-- > 1;) stack.push_back(hitk[i].second);
                st.max_stack = std::max(st.max_stack, (int)stack.size());
                if (!hitk.empty()) cur = hitk[0].second;
                else if (stack.empty()) break;
                else {
                    cur = stack.back();
                    stack.pop_back();
                }
            } else {
                const uint32_t enc = ~(uint32_t)cur, first = enc >> 3, cnt = (enc & 7u) + 1u;
                st.leaves++;
                for (uint32_t i = first; i < first + cnt; ++i) {
                    st.tris++;
                    L2.access(tri_base + i); // padded record: one line per triangle
                    double t;
                    if (tri_hit(tris[B.order[i]], r, 1e-4, tbest, t)) tbest = t;
                }
                if (stack.empty()) break;
                cur = stack.back();
                stack.pop_back();
            }
        }
    }
    const double n = (double)(rays.size() - warm);
    st.visits /= n;
    st.tris /= n;
    st.leaves /= n;
    st.lines = (double)L2.misses / n;
    st.node_lines = (double)node_misses / n;
    return st;
}

// children consecutive, depth-first "first child next" (the product's numbering)
static void renumber(std::vector<WNode>& nodes) {
    std::vector<WNode> out;
    out.reserve(nodes.size());
    struct Open { int32_t old_i; int32_t slot; };
    std::vector<Open> todo;
    out.emplace_back();
    todo.push_back({0, 0});
    while (!todo.empty()) {
        const Open o = todo.back();
        todo.pop_back();
        WNode n = nodes[o.old_i];
        int32_t olds[16];
        for (int i = 0; i < n.nk; ++i) {
            olds[i] = n.ref[i];
            if (n.ref[i] >= 0) {
                n.ref[i] = (int32_t)out.size();
                out.emplace_back();
            }
        }
        for (int i = n.nk - 1; i >= 0; --i)
            if (olds[i] >= 0) todo.push_back({olds[i], n.ref[i]});
        out[o.slot] = n;
    }
    nodes.swap(out);
}

static double area(const WBox& b) {
    const double x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2];
    return x * y + y * z + z * x;
}

// widen: a node absorbs its inner child with the largest box while the children fit
static std::vector<WNode> widen(const std::vector<WNode>& in, int width) {
    std::vector<WNode> out(in.size());
    std::vector<char> used(in.size(), 0);
    std::vector<int32_t> todo{0};
    while (!todo.empty()) {
        const int32_t i = todo.back();
        todo.pop_back();
        WNode n = in[i];
        for (;;) {
            int best = -1;
            double ba = -1;
            for (int k = 0; k < n.nk; ++k)
                if (n.ref[k] >= 0 && n.nk - 1 + in[n.ref[k]].nk <= width && area(n.box[k]) > ba) {
                    ba = area(n.box[k]);
                    best = k;
                }
            if (best < 0) break;
            const WNode& c = in[n.ref[best]];
            n.box[best] = c.box[0];
            n.ref[best] = c.ref[0];
            for (int k = 1; k < c.nk; ++k) {
                n.box[n.nk] = c.box[k];
                n.ref[n.nk] = c.ref[k];
                n.nk++;
            }
        }
        out[i] = n;
        used[i] = 1;
        for (int k = 0; k < n.nk; ++k)
            if (n.ref[k] >= 0) todo.push_back(n.ref[k]);
    }
    renumber(out);
    return out;
}

// child boxes on a per-node 8-bit grid (power-of-two steps of the global 16-bit grid), rounded outward
static void quantise8(std::vector<WNode>& nodes, const float g0[3], const float gs[3], int bits = 8) {
    const double top = (double)((1 << bits) - 1);
    for (WNode& n : nodes) {
        for (int a = 0; a < 3; ++a) {
            double lo = 1e300, hi = -1e300;
            for (int k = 0; k < n.nk; ++k) {
                lo = std::min(lo, (double)n.box[k].lo[a]);
                hi = std::max(hi, (double)n.box[k].hi[a]);
            }
            const double qlo = std::floor((lo - g0[a]) / gs[a]), qhi = std::ceil((hi - g0[a]) / gs[a]);
            int e = 0;
            while ((qhi - qlo) > top * std::ldexp(1.0, e)) ++e;
            const double step = std::ldexp(1.0, e) * gs[a];
            const double org = g0[a] + qlo * gs[a];
            for (int k = 0; k < n.nk; ++k) {
                const double l8 = std::floor((n.box[k].lo[a] - org) / step), h8 = std::ceil((n.box[k].hi[a] - org) / step);
                n.box[k].lo[a] = (float)(org + std::max(0.0, l8) * step);
                n.box[k].hi[a] = (float)(org + std::min(top, std::max(h8, l8 + 1)) * step);
                n.box[k].lo[a] = std::nextafter(n.box[k].lo[a], -1e30f);
                n.box[k].hi[a] = std::nextafter(n.box[k].hi[a], 1e30f);
            }
        }
    }
}

int main(int argc, char** argv) {
    const size_t n_tris = argc > 1 ? (size_t)std::atoll(argv[1]) : 8000000;
    const size_t n_rays = argc > 2 ? (size_t)std::atoll(argv[2]) : 300000;
    std::mt19937_64 rng(4);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    std::vector<HostTri> tris(n_tris);
    for (size_t i = 0; i < n_tris; ++i) {
        HostTri& T = tris[i];
        double c[3] = {U(rng), U(rng), U(rng)};
        for (int a = 0; a < 3; ++a) {
            T.lo[a] = 1e300;
            T.hi[a] = -1e300;
        }
        for (int v = 0; v < 3; ++v)
            for (int a = 0; a < 3; ++a) {
                T.v[v][a] = c[a] + (U(rng) - 0.5) * 0.01;
                T.lo[a] = std::min(T.lo[a], T.v[v][a]);
                T.hi[a] = std::max(T.hi[a], T.v[v][a]);
            }
        for (int a = 0; a < 3; ++a)
            if (T.hi[a] - T.lo[a] < 1e-4) {
                T.lo[a] -= 5e-5;
                T.hi[a] += 5e-5;
            }
        T.material = 0;
        T.prim = (int32_t)i;
    }
    BuiltBVH B;
    std::string err;
    if (!build_bvh(tris, B, &err)) {
        std::fprintf(stderr, "build failed: %s\n", err.c_str());
        return 1;
    }
    std::printf("built: %zu nodes (4-wide, stack need %d)\n", B.nodes.size(), B.stack_need);
    std::fflush(stdout);
    std::vector<Ray> rays(n_rays);
    for (Ray& r : rays) {
        for (int a = 0; a < 3; ++a) r.o[a] = U(rng);
        const double z = 2 * U(rng) - 1, phi = 6.283185307179586 * U(rng), s = std::sqrt(std::max(0.0, 1 - z * z));
        r.d[0] = s * std::cos(phi);
        r.d[1] = s * std::sin(phi);
        r.d[2] = z;
    }
    // the product's tree as generic nodes
    Variant v4{"4-wide, 64 B, 16-bit boxes (product)", {}, 64, 1};
    v4.nodes.resize(B.nodes.size());
    for (size_t i = 0; i < B.nodes.size(); ++i) {
        const DNode& d = B.nodes[i];
        WNode& w = v4.nodes[i];
        for (int k = 0; k < 4; ++k) {
            if (d.ref[k] == (int32_t)0x80000000) continue;
            const uint32_t q[3] = {d.bx[k], d.by[k], d.bz[k]};
            for (int a = 0; a < 3; ++a) {
                w.box[w.nk].lo[a] = B.grid_origin[a] + (float)(q[a] & 0xffffu) * B.grid_step[a];
                w.box[w.nk].hi[a] = B.grid_origin[a] + (float)(q[a] >> 16) * B.grid_step[a];
            }
            w.ref[w.nk] = d.ref[k];
            w.nk++;
        }
    }
    std::vector<Variant> vs;
    vs.push_back(v4);
    {
        Variant v{"4-wide, 32 B, 8-bit boxes (4 per line)", v4.nodes, 32, 1};
        quantise8(v.nodes, B.grid_origin, B.grid_step);
        vs.push_back(v);
    }
    {
        Variant v{"4-wide, 32 B, 6-bit boxes (4 per line)", v4.nodes, 32, 1};
        quantise8(v.nodes, B.grid_origin, B.grid_step, 6);
        vs.push_back(v);
        Variant w{"4-wide, 32 B, 7-bit boxes (4 per line)", v4.nodes, 32, 1};
        quantise8(w.nodes, B.grid_origin, B.grid_step, 7);
        vs.push_back(w);
    }
    {
        Variant v{"8-wide, 128 B, 16-bit boxes", widen(v4.nodes, 8), 128, 1};
        vs.push_back(v);
        Variant c{"8-wide, 64 B, 8-bit boxes (2 per line)", v.nodes, 64, 1};
        quantise8(c.nodes, B.grid_origin, B.grid_step);
        vs.push_back(c);
    }
    {
        Variant v{"6-wide, 64 B, 8-bit boxes", widen(v4.nodes, 6), 64, 1};
        quantise8(v.nodes, B.grid_origin, B.grid_step);
        vs.push_back(v);
    }
    {
        Variant v{"16-wide, 128 B, 8-bit boxes", widen(v4.nodes, 16), 128, 1};
        quantise8(v.nodes, B.grid_origin, B.grid_step);
        vs.push_back(v);
    }
    std::vector<Stats> res(vs.size());
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < vs.size(); ++i) res[i] = run(vs[i], B, tris, rays, n_rays / 3);
    for (size_t i = 0; i < vs.size(); ++i) {
        double avg = 0;
        for (const WNode& n : vs[i].nodes) avg += n.nk;
        std::printf("%-42s nodes %8zu (%.1f MB, %.2f children)  visits/ray %.1f  leaves %.2f  tri tests %.2f  L2-miss lines/ray %.1f (nodes %.1f)  max stack %d\n",
                    vs[i].name, vs[i].nodes.size(), vs[i].nodes.size() * (double)vs[i].node_bytes / 1e6, avg / vs[i].nodes.size(), res[i].visits, res[i].leaves,
                    res[i].tris, res[i].lines, res[i].node_lines, res[i].max_stack);
    }
    return 0;
}

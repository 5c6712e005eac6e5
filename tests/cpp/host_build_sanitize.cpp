// Test helper (built with -fsanitize=address,undefined): the host-side scene preparation of libprt_hip —
// Triangle precompute, binned-SAH BVH with 16-bit quantised boxes, light tree — on random and degenerate
// input, with the tree invariants checked: every triangle in exactly one leaf, every leaf's triangles inside
// its (dequantised) box, child boxes inside the parent's, depth within the traversal stack.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../pooraytracer_amd/csrc/prt_host.h"

using namespace prt;

static int fail(const char* what) {
    std::fprintf(stderr, "FAILED: %s\n", what);
    return 1;
}

struct Box {
    double lo[3], hi[3];
};

static int check_tree(const BuiltBVH& b, const std::vector<HostTri>& tris) {
    const size_t n = tris.size();
    std::vector<int> seen(n, 0);
    auto deq = [&](const uint16_t q[2], int a, double& lo, double& hi) {
        lo = (double)b.grid_origin[a] + (double)q[0] * (double)b.grid_step[a];
        hi = (double)b.grid_origin[a] + (double)q[1] * (double)b.grid_step[a];
    };
    struct Item { int32_t ref; Box box; uint32_t depth; };
    std::vector<Item> stack;
    Box all;
    for (int a = 0; a < 3; ++a) { all.lo[a] = -1e300; all.hi[a] = 1e300; }
    stack.push_back({0, all, 0});
    uint32_t max_depth = 0;
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (it.depth > max_depth) max_depth = it.depth;
        if (it.ref < 0) {
            const uint32_t enc = ~(uint32_t)it.ref, first = enc >> 3, cnt = (enc & 7u) + 1u;
            if (cnt > PRT_LEAF_MAX || first + cnt > n) return fail("leaf range");
            for (uint32_t i = first; i < first + cnt; ++i) {
                const HostTri& T = tris[b.order[i]];
                seen[b.order[i]]++;
                for (int a = 0; a < 3; ++a)
                    if (T.lo[a] < it.box.lo[a] || T.hi[a] > it.box.hi[a]) return fail("triangle outside its leaf box");
            }
            continue;
        }
        if ((size_t)it.ref >= b.nodes.size()) return fail("node index");
        const DNode& nd = b.nodes[it.ref];
        int kids = 0;
        for (int i = 0; i < 4; ++i) {
            if (nd.ref[i] == (int32_t)0x80000000) {
                if (nd.bx[i] != 0x0000ffffu || nd.by[i] != 0x0000ffffu || nd.bz[i] != 0x0000ffffu) return fail("unused slot is not inverted");
                continue;
            }
            if (i != kids) return fail("unused slot before a used one"); // the traversal checks the refs of slots 2 and 3 only
            ++kids;
            if (n == 1 && i == 1 && nd.ref[1] == nd.ref[0]) continue; // the one-triangle root lists its leaf twice (like the reference's span-1 node)
            Box c;
            const uint32_t w[3] = {nd.bx[i], nd.by[i], nd.bz[i]};
            for (int a = 0; a < 3; ++a) {
                const uint16_t q[2] = {(uint16_t)(w[a] & 0xffffu), (uint16_t)(w[a] >> 16)};
                deq(q, a, c.lo[a], c.hi[a]);
                // a child box may stick out of its parent's by the outward rounding of one grid step, never more
                if (c.lo[a] < it.box.lo[a] - 1.0001 * b.grid_step[a] || c.hi[a] > it.box.hi[a] + 1.0001 * b.grid_step[a]) return fail("child box outside its parent");
            }
            stack.push_back({nd.ref[i], c, it.depth + 1});
        }
        if (kids < 2) return fail("wide node with fewer than two children");
        continue;
    }
    for (size_t i = 0; i < n; ++i)
        if (seen[i] != 1) return fail("triangle not in exactly one leaf");
    // worst-case stack use of a traversal: entering a node with k children leaves up to k-1 siblings on the stack
    std::vector<int> need(b.nodes.size(), 0);
    for (size_t i = b.nodes.size(); i-- > 0;) { // children have larger indices than their parent
        int k = 0, worst = 0;
        for (int c = 0; c < 4; ++c) {
            if (b.nodes[i].ref[c] == (int32_t)0x80000000) continue;
            ++k;
            if (b.nodes[i].ref[c] >= 0) {
                if ((size_t)b.nodes[i].ref[c] <= i) return fail("child index not after its parent");
                worst = std::max(worst, need[b.nodes[i].ref[c]]);
            }
        }
        need[i] = k - 1 + worst;
    }
    if (!need.empty() && need[0] > PRT_STACK_DEPTH) return fail("a traversal could overflow the stack");
    (void)max_depth;
    return 0;
}

int main() {
    std::mt19937_64 rng(99);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    int n_shallow = 0;
    for (int round = 0; round < 7; ++round) {
        // rounds 5 and 6: the scene far from the world origin (one fp32 ulp there = 1/16 resp. 2 scene units: the builders work on
        // boxes relative to the grid origin, so the tree invariants below must hold exactly as they do at the origin)
        const size_t n = round == 0 ? 1 : round == 1 ? 2 : round == 2 ? 37 : round == 3 ? 5000 : round == 4 ? 60000 : 5000;
        const double off[3] = {round == 5 ? 1.0e6 : round == 6 ? -3.0e7 : 0.0, round == 5 ? -2.0e6 : round == 6 ? 1.0e7 : 0.0, round >= 5 ? 3.0e6 : 0.0};
        std::vector<double> v(n * 9), uv(n * 6), nr(n * 9, 0.0);
        for (size_t t = 0; t < n; ++t) {
            const double cx = U(rng) * 10 - 5, cy = U(rng) * 2, cz = U(rng) * 10 - 5;
            const double s = (t % 97 == 0) ? 3.0 : 0.02 + 0.1 * U(rng); // a few large triangles among small ones
            for (int k = 0; k < 3; ++k) {
                v[t * 9 + k * 3 + 0] = off[0] + cx + s * (U(rng) - 0.5);
                v[t * 9 + k * 3 + 1] = off[1] + cy + s * (U(rng) - 0.5);
                v[t * 9 + k * 3 + 2] = off[2] + cz + s * (U(rng) - 0.5);
                uv[t * 6 + k * 2] = U(rng);
                uv[t * 6 + k * 2 + 1] = U(rng);
            }
            if (t % 211 == 5) // degenerate: all three vertices equal (NaN normal -> fallbacks)
                for (int k = 1; k < 3; ++k)
                    for (int a = 0; a < 3; ++a) v[t * 9 + k * 3 + a] = v[t * 9 + a];
            if (t % 223 == 7) // degenerate texture coordinates (tangent fallback)
                for (int k = 0; k < 3; ++k) uv[t * 6 + k * 2] = uv[t * 6 + k * 2 + 1] = 0.25;
        }
        PrtMaterial mats[2] = {};
        mats[0].type = PRT_MAT_LAMBERTIAN; mats[0].texture = -1; mats[0].kd[0] = mats[0].kd[1] = mats[0].kd[2] = 0.5;
        mats[1].type = PRT_MAT_DIFFUSE_LIGHT; mats[1].texture = -1; mats[1].emission[0] = 5;
        const size_t n_light = n >= 37 ? n / 10 : 0;
        // lights: one emissive mesh (a span-1 top node: the wrap-around case), or — last round — three of unequal size (the
        // lights BVH over them then holds a span-1 node INSIDE: BVH.cpp:21-23 with span 3 -> 1 + 2)
        const bool three = round == 4;
        const size_t l0 = n - n_light, l1 = l0 + n_light / 2, l2 = l1 + n_light / 3;
        const uint64_t first3[5] = {0, l0, l1, l2, n};
        const int32_t mm3[4] = {0, 1, 1, 1};
        const uint64_t first[3] = {0, n - n_light, n};
        const int32_t mm[2] = {0, 1};
        PrtSceneDesc d = {};
        d.n_tris = n; d.vertices = v.data(); d.normals = nr.data(); d.texcoords = uv.data();
        d.n_meshes = n_light ? 2 : 1; d.n_materials = 2; d.mesh_first_tri = first; d.mesh_material = mm; d.materials = mats;
        const uint64_t first1[2] = {0, n};
        if (!n_light) d.mesh_first_tri = first1;
        if (three) {
            d.n_meshes = 4;
            d.mesh_first_tri = first3;
            d.mesh_material = mm3;
        }
        std::vector<HostTri> tris;
        setup_triangles(d, tris);
        std::vector<DMaterial> dm;
        setup_materials(d, dm);
        LightTree lt;
        build_light_tree(d, tris, dm, lt);
        if (lt.tris.size() != n_light) return fail("light triangle count");
        const bool root_is_table = lt.root >= 0 && (lt.root & PRT_LIGHT_TABLE_BIT);
        if (n_light && !root_is_table && lt.root != 0 && lt.nodes.size() > 0) return fail("light tree root is not node 0 after renumbering");
        auto bad_ref = [&](int32_t r) {
            if (r < 0) return (size_t)(~r) >= lt.tris.size();
            if (r & PRT_LIGHT_TABLE_BIT) return (uint32_t)(r & ~PRT_LIGHT_TABLE_BIT) >= lt.n_tables;
            return (size_t)r >= lt.nodes.size();
        };
        for (const DLightNode& ln : lt.nodes)
            if (bad_ref(ln.left) || bad_ref(ln.right)) return fail("light tree reference out of range");
        if (n_light >= PRT_LIGHT_TABLE_MIN) {
            // O(1) light pick: tables exist (one per emissive mesh), the nodes the kernels still descend are the few above them,
            // and the pick equals the full tree's descent — at every threshold and its neighbours, over a window of EVERY float
            // around the total area (p = sqrt(xi) * area rounds up to it: the wrap-around of a span-1 node), and on 2e6 random p
            if (lt.n_tables != (three ? 2u : 1u)) return fail("light tables: one per maximal span-1-free subtree expected (one mesh: 1; three: the lone mesh and the pair)");
            if (lt.nodes.size() > 8) return fail("light tables: the top part should be a handful of nodes");
            auto fbits = [](float f) { uint32_t b; std::memcpy(&b, &f, 4); return b; };
            auto bfloat = [](uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; };
            size_t checked = 0;
            auto same = [&](float p) { ++checked; return light_pick(lt, p) == light_pick_full_tree(lt, p); };
            for (uint32_t t = 0; t < lt.n_tables; ++t) {
                DLightTable h;
                std::memcpy(&h, lt.tab.data() + 8 * (size_t)t, sizeof(h));
                for (uint32_t i = 0; i <= h.n; ++i) {
                    const uint32_t b = lt.tab[h.thr_off + i];
                    for (int dlt = -2; dlt <= 2; ++dlt) {
                        const uint32_t q = b + (uint32_t)dlt;
                        if (q <= 0x7f7fffffu && !same(bfloat(q))) return fail("light pick differs from the descent next to a threshold");
                    }
                }
            }
            const uint32_t tot = fbits((float)lt.area);
            for (uint32_t b = tot - 5000; b <= tot + 5000; ++b)
                if (!same(bfloat(b))) return fail("light pick differs from the descent near the total area (wrap-around)");
            std::uniform_int_distribution<uint32_t> P(0, tot + 1000);
            for (int k = 0; k < 2000000; ++k)
                if (!same(bfloat(P(rng)))) return fail("light pick differs from the descent at a random p");
            // the descent's leaf moves right with p on every table's subtree: sampled monotonicity of the full pick below the wrap
            int32_t last = -1;
            for (uint32_t b = 0; b < tot; b += 257) {
                const int32_t leaf = light_pick_full_tree(lt, bfloat(b));
                if (!three && leaf < last) return fail("descent not monotone on a one-mesh lights list below the total area");
                last = leaf;
            }
            std::printf("light tables: %u tables, %zu top nodes, %zu words, %zu picks compared\n", lt.n_tables, lt.nodes.size(), lt.tab.size(), checked);
        }
        BuiltBVH b;
        std::string err;
        if (!build_bvh(tris, b, &err)) return fail(err.c_str());
        if (b.order.size() != n) return fail("order size");
        if (check_tree(b, tris)) return 1;
        if (!validate_nodes(b.nodes.data(), b.nodes.size(), n, &err)) return fail(err.c_str());
        // the second collapse (small LDS stacks, fp32 render kernels): present exactly when the first one can need more
        // than PRT_STACK_SHALLOW entries, a well-formed tree over the same leaves, and within that bound itself
        if (b.stack_need != tree_stack_need(b.nodes.data(), b.nodes.size())) return fail("stack_need is not the tree's");
        if ((b.stack_need > PRT_STACK_SHALLOW) != !b.nodes_shallow.empty()) return fail("shallow tree present / absent wrongly");
        if (!b.nodes_shallow.empty()) {
            BuiltBVH sh = b;
            sh.nodes = b.nodes_shallow;
            if (check_tree(sh, tris)) return fail("shallow tree fails the structural check");
            if (!validate_nodes(sh.nodes.data(), sh.nodes.size(), n, &err)) return fail(err.c_str());
            if (tree_stack_need(sh.nodes.data(), sh.nodes.size()) > PRT_STACK_SHALLOW) return fail("shallow tree needs more than its budget");
            ++n_shallow;
        }
        std::vector<double> moved(v);
        for (double& x : moved) x = x * 1.5 + 0.25;
        update_triangles(moved.data(), nullptr, tris);
        if (!build_bvh(tris, b, &err) || check_tree(b, tris)) return fail("rebuild after update_triangles");
        std::printf("n=%zu nodes=%zu depth=%u need=%d shallow=%zu lights=%zu ok\n", n, b.nodes.size(), b.depth, b.stack_need, b.nodes_shallow.size(), lt.tris.size());
    }
    return 0;
}

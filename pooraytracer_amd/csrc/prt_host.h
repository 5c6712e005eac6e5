// prt_host.h — host-side scene preparation for libprt_hip.so (no device code here).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/prt.h"
#include "prt_types.h"

namespace prt {

// Result of the Triangle constructor precompute (reference Source/Triangle.cpp:11-53).
struct HostTri {
    double v[3][3];
    double e0[3], e1[3];
    double uv[3][2];
    double normal[3], tangent[3];
    double area, D, w[3];
    double lo[3], hi[3]; // bbox, padded to >= 1e-4 per axis (AABB.cpp:76-82)
    int32_t material, prim;
};

struct BuiltBVH {
    std::vector<DNode> nodes;    // nodes[0] is the root
    std::vector<DNode> nodes_shallow; // 4-wide trees that can need more than PRT_STACK_SHALLOW stack entries: the same
                                      // binary tree collapsed with that budget (same leaves, same triangle order); else empty
    int stack_need = 0;          // stack entries a traversal of `nodes` can need at most
    std::vector<uint32_t> order; // BVH leaf order -> index into the HostTri array
    uint32_t depth = 0;
    float coord_scale = 1.0f;    // >= |every box coordinate| (relative to grid_origin)
    float grid_origin[3] = {0, 0, 0}, grid_step[3] = {1, 1, 1}; // quantisation grid of the 16-bit boxes
    // tools/sim_oct8.cpp only: the binary SAH tree the wide nodes were collapsed from (pre-order; refs as in DNode)
    struct BinNode {
        float lo[2][3], hi[2][3];
        int32_t ref[2];
    };
    bool keep_binary = false;
    std::vector<BinNode> binary;
};

// fp32 box of one triangle for the BVH builders: HostTri::lo/hi RELATIVE to `origin` (an fp32 point just below every
// coordinate of the scene, which becomes the grid origin of the 16-bit nodes), rounded outward + a small absolute inflation
struct PrimBox {
    float lo[3], hi[3];
};
void prim_boxes(const std::vector<HostTri>& tris, std::vector<PrimBox>& out, float origin[3]);
// quantisation grid of the 16-bit boxes over the root box (origin rounds down, 65535 steps reach past hi).
void quant_grid(const float root_lo[3], const float root_hi[3], bool empty, float origin[3], float step[3]);

// BVH built on the device (bvh_build_gpu.hip): nodes and the triangle permutation stay in HBM.
struct DeviceBVH {
    DNode* d_nodes = nullptr;    // hipMalloc'ed, n_nodes entries (caller owns)
    uint32_t* d_order = nullptr; // hipMalloc'ed, n entries: BVH leaf order -> index into the HostTri array
    uint32_t n_nodes = 0, depth = 0;
    float coord_scale = 1.0f;
    float grid_origin[3] = {0, 0, 0}, grid_step[3] = {1, 1, 1};
    double ms_sort = 0, ms_tree = 0, ms_split = 0, ms_total = 0; // hipEvent times of the build phases
};

struct LightTree {
    std::vector<DLightNode> nodes;   // what the kernels descend: the full tree, or (with tables) the part above the tables
    std::vector<DLightTri> tris;     // in area-CDF (leaf) order
    int32_t root = -1;               // node index, or a table ref (PRT_LIGHT_TABLE_BIT)
    double area = 0.0;
    std::vector<uint32_t> tab;       // DLightTable headers + thresholds + buckets (prt_types.h); empty: no tables
    uint32_t n_tables = 0;
    std::vector<DLightNode> full_nodes; // the full tree (tests, verification)
    int32_t full_root = -1;
};
// Host twin of the kernels' light pick (prt_device.h, sample_lights): leaf (CDF index) reached by the float p.
int32_t light_pick(const LightTree& lt, float p);
int32_t light_pick_full_tree(const LightTree& lt, float p);

// Triangle.cpp:11-53 for every triangle of the description.
void setup_triangles(const PrtSceneDesc& d, std::vector<HostTri>& out);
// Re-runs the Triangle constructor for new vertex positions (topology, materials, uv unchanged).
void update_triangles(const double* vertices, const double* normals, std::vector<HostTri>& tris);
// Material table incl. SetProbabilitiesByNs / HasEmission / SkipLightSampling (Material.h).
void setup_materials(const PrtSceneDesc& d, std::vector<DMaterial>& out);
// Binned-SAH BVH2, depth-bounded to PRT_STACK_DEPTH, child boxes rounded outward to fp32.
// Returns false (with *err set) if a compiled-in limit is exceeded.
bool build_bvh(const std::vector<HostTri>& tris, BuiltBVH& out, std::string* err);
// Structural check of a flattened tree (refs in range, every triangle in exactly one leaf, stack bound).
bool validate_nodes(const DNode* nodes, size_t n_nodes, size_t n_tris, std::string* err);
// Largest number of stack entries a traversal of this tree can need (<= PRT_STACK_DEPTH for a builder's tree).
int tree_stack_need(const DNode* nodes, size_t n_nodes);
// Same tree family built on the current HIP device from the same fp32 boxes (n >= 2): Morton sort, box
// segment tree, level-synchronous SAH splits along the Morton order.  Returns false with *err set.
bool build_bvh_device(const PrimBox* h_boxes, const float box_origin[3], size_t n, DeviceBVH& out, std::string* err);
// The reference's lights object graph (main.cpp:36-45, BVH.cpp:7-48) reduced to what
// BVHNode::Sample/TraverseSample read: per-node left area + children, leaves in CDF order.
void build_light_tree(const PrtSceneDesc& d, const std::vector<HostTri>& tris, const std::vector<DMaterial>& mats,
                      LightTree& out);
// Camera::Initialize (Camera.cpp:75-106).
void setup_camera(const PrtCamera& c, DCamera& out);

// Host twin of the device's seed_key() (prt_device.h, Rng): the seed of a launch is hashed once, here.
inline uint64_t seed_key(uint64_t seed) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ULL;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

} // namespace prt

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
    std::vector<uint32_t> order; // BVH leaf order -> index into the HostTri array
    uint32_t depth = 0;
    float coord_scale = 1.0f;    // >= |every box coordinate|
    float grid_origin[3] = {0, 0, 0}, grid_step[3] = {1, 1, 1}; // PRT_NODE16 quantisation grid
};

struct LightTree {
    std::vector<DLightNode> nodes;
    std::vector<DLightTri> tris; // in area-CDF (leaf) order
    int32_t root = -1;
    double area = 0.0;
};

// Triangle.cpp:11-53 for every triangle of the description.
void setup_triangles(const PrtSceneDesc& d, std::vector<HostTri>& out);
// Material table incl. SetProbabilitiesByNs / HasEmission / SkipLightSampling (Material.h).
void setup_materials(const PrtSceneDesc& d, std::vector<DMaterial>& out);
// Binned-SAH BVH2, depth-bounded to PRT_STACK_DEPTH, child boxes rounded outward to fp32.
// Returns false (with *err set) if a compiled-in limit is exceeded.
bool build_bvh(const std::vector<HostTri>& tris, BuiltBVH& out, std::string* err);
// The reference's lights object graph (main.cpp:36-45, BVH.cpp:7-48) reduced to what
// BVHNode::Sample/TraverseSample read: per-node left area + children, leaves in CDF order.
void build_light_tree(const PrtSceneDesc& d, const std::vector<HostTri>& tris, const std::vector<DMaterial>& mats,
                      LightTree& out);
// Camera::Initialize (Camera.cpp:75-106).
void setup_camera(const PrtCamera& c, DCamera& out);

} // namespace prt

// Triangle.h — mirror of Source/Triangle.h:12-43 (public geometry fields, Mesh).  The constructor
// precompute that feeds the intersection test (normal, tangent, D, w, bbox: Triangle.cpp:11-53) is
// done by libprt_hip at scene creation; the host class keeps the inputs plus area and bbox.
#pragma once
#include <array>
#include <string>

#include "Hittable.h"
#include "HittableList.h"

namespace Pooraytracer {
class Triangle : public Hittable {
public:
    Triangle(const std::array<vec3, 3>& vertices, const std::array<vec3, 3>& normals, const std::array<vec2, 3>& texCoords,
             std::shared_ptr<Material> material);
    AABB BoundingBox() const override { return bbox; }
    double GetArea() const override { return area; }
    void Flatten(SceneFlattener& out) const override;

public:
    std::array<vec3, 3> vertices; // v0, v1, v2, right-handed
    std::array<vec3, 2> edges;    // e0: v1-v0, e1: v2-v0
    std::array<vec2, 3> texCoords;
    std::array<vec3, 3> vertexNormals; // degenerate-face fallback only (Triangle.cpp:21-29)
    vec3 normal;
    vec3 tangent;
    double area;
    AABB bbox;
    std::shared_ptr<Material> material;
};

class Mesh : public HittableList {
public:
    Mesh() = default;
    Mesh(const std::string& name, const std::vector<std::shared_ptr<Hittable>>& triangles, shared_ptr<Material> material);
    void Flatten(SceneFlattener& out) const override;

public:
    std::string name;
    std::shared_ptr<Material> material;
};
} // namespace Pooraytracer

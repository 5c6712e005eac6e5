// BVH.h — mirror of Source/BVH.h:10-35.  The reference's BVHNode is a pointer tree built by median
// split (BVH.cpp:7-48); here it is a grouping node: the traversal BVH (binned SAH, flattened, fp32
// conservative boxes) is built by libprt_hip from the flat triangle list, and the reference's own
// tree shape is replayed inside the library only where it is observable — the area-CDF order of the
// light triangles (BVH.cpp:86-100).  Construction does NOT reorder mesh->objects.
#pragma once
#include "AABB.h"
#include "Hittable.h"
#include "HittableList.h"
#include "Triangle.h"

namespace Pooraytracer {
class BVHNode : public Hittable {
public:
    BVHNode(HittableList list);
    BVHNode(shared_ptr<Mesh> mesh);
    BVHNode(std::vector<shared_ptr<Hittable>>& objects, size_t start, size_t end);
    AABB BoundingBox() const override { return bbox; }
    double GetArea() const override { return area; }
    void Flatten(SceneFlattener& out) const override;

public:
    AABB bbox;

private:
    shared_ptr<Mesh> mesh_;                      // set by the Mesh constructor: keeps the mesh boundary
    std::vector<shared_ptr<Hittable>> objects_;  // otherwise: the span it was built over
    double area = 0.0;
    void Init();
};
} // namespace Pooraytracer

// HittableList.h — mirror of Source/HittableList.h:12-63.
#pragma once
#include <memory>
#include <vector>

#include "Hittable.h"

namespace Pooraytracer {
using std::make_shared;
using std::shared_ptr;

class HittableList : public Hittable {
public:
    std::vector<shared_ptr<Hittable>> objects;
    HittableList() = default;
    HittableList(shared_ptr<Hittable> object) { Add(object); }
    void Clear() { objects.clear(); }
    void Add(shared_ptr<Hittable> object) {
        objects.push_back(object);
        area += object->GetArea();
        bbox = AABB(bbox, object->BoundingBox());
    }
    AABB BoundingBox() const override { return bbox; }
    double GetArea() const override { return area; }
    void Flatten(SceneFlattener& out) const override;

public:
    AABB bbox;
    double area = 0.0;
};
} // namespace Pooraytracer

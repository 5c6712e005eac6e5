// Hittable.h — mirror of Source/Hittable.h:17-38.
//
// On the host the scene-graph classes are DESCRIPTORS: they carry the same public data and
// constructors as the reference's, and Camera::Render flattens them into a PrtSceneDesc for the HIP
// library (include/prt.h).  Intersection itself never runs on the host: the base Hittable::Hit /
// Sample below forward single queries to the device through the same C ABI (K1 / k_sample_lights).
#pragma once
#include <memory>

#include "AABB.h"
#include "Math.h"

namespace Pooraytracer {

class Ray;
class Material;
class SceneFlattener;

class HitRecord {
public:
    vec3 position;
    double time = 0;
    vec3 normal; // on the same side as the ray
    vec3 tangent;
    vec2 uv;
    std::shared_ptr<Material> material;
    bool bFrontFace = false;
    void SetFaceNormal(const Ray& ray, const vec3& outwordNormal);
};

class Hittable {
public:
    virtual ~Hittable();
    // world.Hit(ray, domain, record): one-ray batch through prt_trace_closest (device).
    virtual bool Hit(const Ray& ray, Interval domain, HitRecord& record) const;
    virtual AABB BoundingBox() const = 0;
    // lights.Sample(origin, record, pdf): one sample through prt_sample_lights (device).
    virtual void Sample(const point3& origin, HitRecord& samplePointRecord, double& pdf) const;
    virtual double GetArea() const { return 0.0; }
    // Appends this object's meshes/triangles, in construction order, to the flat description.
    virtual void Flatten(SceneFlattener& out) const = 0;

private:
    struct DeviceCache;
    mutable std::shared_ptr<DeviceCache> cache_;
    friend class Camera;
    DeviceCache& Device() const;
};

} // namespace Pooraytracer

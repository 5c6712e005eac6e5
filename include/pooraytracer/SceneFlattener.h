// SceneFlattener.h — collects a Hittable object graph into the flat arrays of PrtSceneDesc
// (include/prt.h): meshes in construction order, triangles in mesh order, one material per mesh
// (reference: Mesh, Source/Triangle.h:36-43; one material per shape, Source/Model.cpp:118).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../prt.h"
#include "Math.h"

namespace Pooraytracer {
class Material;
class Texture;
class Triangle;

class SceneFlattener {
public:
    void BeginMesh(const std::string& name, const std::shared_ptr<Material>& material);
    void EndMesh();
    void AddTriangle(const Triangle& t); // outside Begin/EndMesh: grouped into anonymous per-material meshes

    // Fills `desc` with pointers into this object (valid while it lives).
    void Describe(PrtSceneDesc& desc);

    std::vector<double> vertices, normals, texcoords;
    std::vector<uint64_t> meshFirstTri{0};
    std::vector<int32_t> meshMaterial;
    std::vector<std::string> meshNames;
    std::vector<std::shared_ptr<Material>> materials;
    std::vector<const Triangle*> triangles; // PrtHit::prim -> host triangle

private:
    int MaterialIndex(const std::shared_ptr<Material>& m);
    bool inMesh_ = false, looseOpen_ = false;
    std::vector<PrtMaterial> matTable_;
    std::vector<PrtTexture> texTable_;
    std::vector<std::shared_ptr<Texture>> textures_;
};
} // namespace Pooraytracer

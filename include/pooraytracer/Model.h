// Model.h — mirror of Source/Model.h:14-31: loads <dir>/<name>.obj (+ .mtl) and <dir>/<name>.xml and
// produces `meshes` (one Mesh per OBJ shape, one material per shape) exactly as Source/Model.cpp:53-193.
//
// The reference uses tinyobjloader + tinyxml2 + stb_image; none of them is available here, so this is
// a small self-contained reader of the subset the reference consumes:
//   OBJ: v / vt / vn / f (triangles; polygons are fan-triangulated), g / o (new shape), usemtl, mtllib,
//        negative indices; anonymous groups are named Group_<n> in memory (the reference rewrites the
//        .obj file in place, Model.cpp:195-276 — not reproduced);
//   MTL: newmtl, Kd, Ks, Ns, map_Kd, map_Ks;
//   XML: <light mtlname="..." radiance="r,g,b"/> (Model.cpp:332-360);
//   textures: PNG (8-bit grey/grey+alpha/RGB/RGBA, non-interlaced; own inflate, png_decode.cpp) and binary
//        PPM/PGM (P6/P5) — other formats (JPEG, 16-bit, palette, interlaced) load as the reference's
//        "missing texture" (cyan, Texture.cpp:24).
// Material type comes from the reference's name table (Model.cpp:16-51); unknown names are Lambertian.
#pragma once
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "Material.h"
#include "Triangle.h"

namespace Pooraytracer {
class Model {
public:
    Model() = default;
    Model(const std::string& modelDirectory, const std::string& modelName);
    std::vector<std::shared_ptr<Mesh>> meshes;

    static const std::unordered_map<std::string, MaterialType> materialTypeMap;

private:
    std::string modelDirectory;
    std::string modelName;
    std::unordered_map<std::string, color> lightRadianceMap;
    std::unordered_map<std::string, std::shared_ptr<Material>> materialInstances;
    std::unordered_map<std::string, std::shared_ptr<Texture>> imageTextureInstances;
    void InitializeLightsRadiance();
};
} // namespace Pooraytracer

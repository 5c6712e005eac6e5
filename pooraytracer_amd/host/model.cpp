// model.cpp — scene loader of the host API (include/pooraytracer/Model.h): the caller side of the
// hot-path boundary.  Follows Source/Model.cpp:53-193,278-360.
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "pooraytracer/Model.h"

namespace Pooraytracer {

// Source/Model.cpp:16-51
const std::unordered_map<std::string, MaterialType> Model::materialTypeMap = {
    {"material0", MaterialType::PhoneReflectance}, {"material1", MaterialType::PhoneReflectance},
    {"material2", MaterialType::PhoneReflectance}, {"material3", MaterialType::PhoneReflectance},
    {"material4", MaterialType::PhoneReflectance}, {"light1", MaterialType::DiffuseLight},
    {"light2", MaterialType::DiffuseLight},        {"light3", MaterialType::DiffuseLight},
    {"light4", MaterialType::DiffuseLight},        {"DiffuseWhite", MaterialType::Lambertian},
    {"DiffuseBall", MaterialType::Lambertian},     {"DiffuseYellow", MaterialType::Lambertian},
    {"LeftWall", MaterialType::Lambertian},        {"RightWall", MaterialType::Lambertian},
    {"Light", MaterialType::DiffuseLight},         {"Wall", MaterialType::Lambertian},
    {"quad1", MaterialType::Empty},                {"Mirror", MaterialType::PerfectMirror},
    {"StainlessRough", MaterialType::Lambertian},  {"Towel", MaterialType::Lambertian},
    {"BlackWoodLacquer", MaterialType::Lambertian}, {"Wood", MaterialType::Lambertian},
    {"WoodFloor", MaterialType::Lambertian},       {"Label", MaterialType::Lambertian},
    {"RoughGlass", MaterialType::Lambertian},      {"Plastic", MaterialType::Lambertian},
    {"DarkPlastic", MaterialType::Lambertian},     {"Bin", MaterialType::PerfectMirror},
    {"WallRight", MaterialType::Lambertian},       {"DarkBorder", MaterialType::Lambertian},
    {"Trims", MaterialType::Lambertian},           {"Ceramic", MaterialType::Lambertian}};

namespace {

struct MtlRaw { // the tinyobj::material_t fields the reference reads (Model.cpp:278-330)
    std::string name;
    double diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0};
    double shininess = 1.0;
    std::string diffuse_texname, specular_texname;
};

std::string trim(const std::string& s) {
    const size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    const size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

std::vector<MtlRaw> load_mtl(const std::string& path) {
    std::vector<MtlRaw> out;
    std::ifstream f(path);
    std::string line;
    while (std::getline(f, line)) {
        line = trim(line);
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key;
        ss >> key;
        if (key == "newmtl") {
            out.emplace_back();
            out.back().name = trim(line.substr(6));
        } else if (out.empty()) {
            continue;
        } else if (key == "Kd") {
            ss >> out.back().diffuse[0] >> out.back().diffuse[1] >> out.back().diffuse[2];
        } else if (key == "Ks") {
            ss >> out.back().specular[0] >> out.back().specular[1] >> out.back().specular[2];
        } else if (key == "Ns") {
            ss >> out.back().shininess;
        } else if (key == "map_Kd") {
            out.back().diffuse_texname = trim(line.substr(6));
        } else if (key == "map_Ks") {
            out.back().specular_texname = trim(line.substr(6));
        }
    }
    return out;
}

} // namespace
bool load_png(const std::string& path, int& w, int& h, int& channels, std::vector<unsigned char>& pixels); // png_decode.cpp
bool load_jpeg(const std::string& path, int& w, int& h, int& channels, std::vector<unsigned char>& pixels); // jpeg_decode.cpp
namespace {

// ImageTexture(path), Texture.cpp:10-21, for PNG (8-bit, non-interlaced), JPEG (baseline / progressive
// Huffman, grey or colour) and binary PPM/PGM; anything else = failed load (no data), which renders as
// the reference's cyan "missing texture".
std::shared_ptr<Texture> load_texture(const std::string& path) {
    {
        int w = 0, h = 0, c = 0;
        std::vector<unsigned char> px;
        if (load_png(path, w, h, c, px)) return std::make_shared<ImageTexture>(w, h, c, px.data());
        if (load_jpeg(path, w, h, c, px)) return std::make_shared<ImageTexture>(w, h, c, px.data());
    }
    std::ifstream f(path, std::ios::binary);
    std::string magic;
    int w = 0, h = 0, maxv = 0;
    if (f && (f >> magic) && (magic == "P6" || magic == "P5") && (f >> w >> h >> maxv) && maxv == 255 && w > 0 && h > 0) {
        f.get(); // single whitespace after maxval
        const int c = magic == "P6" ? 3 : 1;
        std::vector<unsigned char> px((size_t)w * h * c);
        f.read(reinterpret_cast<char*>(px.data()), (std::streamsize)px.size());
        if (f) return std::make_shared<ImageTexture>(w, h, c, px.data());
    }
    std::fprintf(stderr, "[pooraytracer] Loading Texture: %s Failed!!\n", path.c_str());
    return std::make_shared<ImageTexture>(0, 0, 0, nullptr);
}

struct Idx {
    int v = -1, vt = -1, vn = -1;
};
struct Shape {
    std::string name;
    std::vector<std::array<Idx, 3>> faces;
    std::vector<int> material_ids; // per face
};

int fix_index(long i, size_t n) { // OBJ: 1-based, negative = relative to the end; -1 = absent, -2 = invalid
    if (i > 0) return i - 1 < 0x7fffffffL ? (int)(i - 1) : 0x7fffffff; // range-checked against the arrays when the mesh is built
    if (i < 0) return (long)n + i >= 0 ? (int)((long)n + i) : -2;
    return -1;
}

} // namespace

void Model::InitializeLightsRadiance() { // Model.cpp:332-360
    const std::string xmlFilePath = modelDirectory + "/" + modelName + ".xml";
    std::ifstream f(xmlFilePath);
    if (!f) {
        std::fprintf(stderr, "[pooraytracer] Failed to load XML file: %s\n", xmlFilePath.c_str());
        return;
    }
    std::stringstream buf;
    buf << f.rdbuf();
    const std::string s = buf.str();
    auto attr = [&](size_t b, size_t e, const std::string& name, std::string& out) {
        size_t p = s.find(name + "=", b);
        if (p == std::string::npos || p > e) return false;
        p = s.find_first_of("\"'", p);
        if (p == std::string::npos || p > e) return false;
        const size_t q = s.find(s[p], p + 1);
        if (q == std::string::npos) return false;
        out = s.substr(p + 1, q - p - 1);
        return true;
    };
    size_t pos = 0;
    while ((pos = s.find("<light", pos)) != std::string::npos) {
        const size_t e = s.find('>', pos);
        if (e == std::string::npos) break;
        std::string name, rad;
        if (attr(pos, e, "mtlname", name) && attr(pos, e, "radiance", rad)) {
            double x = 0., y = 0., z = 0.;
            char comma;
            std::stringstream ss(rad);
            ss >> x >> comma >> y >> comma >> z;
            lightRadianceMap[name] = color(x, y, z);
        }
        pos = e;
    }
}

Model::Model(const std::string& dir, const std::string& name) : modelDirectory(dir), modelName(name) {
    const std::string modelPath = modelDirectory + "/" + modelName + ".obj";
    InitializeLightsRadiance();

    std::ifstream f(modelPath);
    if (!f) {
        std::fprintf(stderr, "[pooraytracer] Open & Process Obj File Failed: %s\n", modelPath.c_str());
        return;
    }
    std::vector<double> V, VT, VN;
    std::vector<MtlRaw> materials;
    std::unordered_map<std::string, int> materialIndex;
    std::vector<Shape> shapes;
    Shape cur;
    int curMat = -1;
    int anonymousGroupCount = -1;
    auto flush = [&]() {
        if (!cur.faces.empty()) shapes.push_back(cur);
        cur.faces.clear();
        cur.material_ids.clear();
    };
    std::string line;
    while (std::getline(f, line)) {
        const std::string t = trim(line);
        if (t.empty() || t[0] == '#') continue;
        std::istringstream ss(t);
        std::string key;
        ss >> key;
        if (key == "v") {
            double x, y, z;
            ss >> x >> y >> z;
            V.insert(V.end(), {x, y, z});
        } else if (key == "vt") {
            double u = 0, v = 0;
            ss >> u >> v;
            VT.insert(VT.end(), {u, v});
        } else if (key == "vn") {
            double x, y, z;
            ss >> x >> y >> z;
            VN.insert(VN.end(), {x, y, z});
        } else if (key == "g" || key == "o") {
            flush();
            std::string gname = t.size() > 2 ? trim(t.substr(2)) : "";
            if (gname.empty()) gname = "Group_" + std::to_string(++anonymousGroupCount); // Model.cpp:217-231 (in memory only)
            cur.name = gname;
        } else if (key == "mtllib") {
            for (auto& m : load_mtl(modelDirectory + "/" + trim(t.substr(6)))) {
                if (!materialIndex.count(m.name)) {
                    materialIndex[m.name] = (int)materials.size();
                    materials.push_back(m);
                }
            }
        } else if (key == "usemtl") {
            const std::string mname = trim(t.substr(6));
            auto it = materialIndex.find(mname);
            curMat = it == materialIndex.end() ? -1 : it->second;
        } else if (key == "f") {
            std::vector<Idx> poly;
            std::string tok;
            while (ss >> tok) {
                Idx ix;
                long a = 0, b = 0, c = 0;
                const char* p = tok.c_str();
                char* end;
                a = std::strtol(p, &end, 10);
                if (*end == '/') {
                    p = end + 1;
                    if (*p != '/') b = std::strtol(p, &end, 10);
                    else end = const_cast<char*>(p);
                    if (*end == '/') c = std::strtol(end + 1, &end, 10);
                }
                ix.v = fix_index(a, V.size() / 3);
                ix.vt = fix_index(b, VT.size() / 2);
                ix.vn = fix_index(c, VN.size() / 3);
                poly.push_back(ix);
            }
            for (size_t k = 1; k + 1 < poly.size(); ++k) { // fan triangulation (triangles pass through unchanged)
                cur.faces.push_back({poly[0], poly[k], poly[k + 1]});
                cur.material_ids.push_back(curMat);
            }
        }
    }
    flush();

    // Loading textures (Model.cpp:79-96; the specular branch re-loads the DIFFUSE name, as upstream)
    for (const auto& m : materials) {
        if (!m.diffuse_texname.empty() && !imageTextureInstances.count(m.diffuse_texname))
            imageTextureInstances[m.diffuse_texname] = load_texture(modelDirectory + "/" + m.diffuse_texname);
    }
    // Creating material instances (Model.cpp:99-110, 278-330)
    for (const auto& m : materials) {
        if (materialInstances.count(m.name)) continue;
        MaterialType type = MaterialType::Lambertian;
        auto it = materialTypeMap.find(m.name);
        if (it != materialTypeMap.end()) type = it->second;
        const color Kd(m.diffuse[0], m.diffuse[1], m.diffuse[2]), Ks(m.specular[0], m.specular[1], m.specular[2]);
        std::shared_ptr<Material> inst;
        switch (type) {
        case MaterialType::PhoneReflectance:
            inst = m.diffuse_texname.empty()
                       ? std::make_shared<PhoneReflectance>(Kd, Ks, m.shininess)
                       : std::make_shared<PhoneReflectance>(imageTextureInstances.at(m.diffuse_texname), Ks, m.shininess);
            break;
        case MaterialType::DiffuseLight: inst = std::make_shared<DiffuseLight>(lightRadianceMap.at(m.name)); break; // throws like .at() upstream
        case MaterialType::PerfectMirror: inst = std::make_shared<PerfectMirror>(); break;
        case MaterialType::CookTorrance:
            inst = std::make_shared<CookTorrance>(Kd, 0.3, 0.3, vec3(0.1, 0.5, 1.5), vec3(4.0, 0.02, 0.3)); // Model.cpp:304-314
            break;
        case MaterialType::DebugMaterial: inst = std::make_shared<DebugMaterial>(Kd); break;
        case MaterialType::Empty: inst = std::make_shared<EmptyMaterial>(); break;
        case MaterialType::Lambertian:
        default:
            inst = m.diffuse_texname.empty() ? std::make_shared<Lambertian>(Kd)
                                             : std::make_shared<Lambertian>(imageTextureInstances.at(m.diffuse_texname));
            break;
        }
        materialInstances[m.name] = inst;
    }

    // Shapes -> meshes (Model.cpp:115-191): the FIRST face's material is the shape's material
    for (const Shape& sh : shapes) {
        const int mi = sh.material_ids[0];
        if (mi < 0) throw std::runtime_error("shape '" + sh.name + "' has no material (usemtl missing)");
        std::shared_ptr<Material> material = materialInstances.at(materials[mi].name);
        std::vector<std::shared_ptr<Hittable>> tris;
        for (const auto& face : sh.faces) {
            std::array<vec3, 3> vs, ns;
            std::array<vec2, 3> uv;
            for (int k = 0; k < 3; ++k) {
                const Idx& ix = face[k];
                // indices come from the file: a face may name a vertex / normal / texcoord that does not exist
                if (ix.v < 0 || (size_t)ix.v >= V.size() / 3 || (ix.vn >= 0 && (size_t)ix.vn >= VN.size() / 3) ||
                    (ix.vt >= 0 && (size_t)ix.vt >= VT.size() / 2))
                    throw std::runtime_error("shape '" + sh.name + "': face index out of range");
                vs[k] = vec3(V[3 * ix.v], V[3 * ix.v + 1], V[3 * ix.v + 2]);
                ns[k] = ix.vn >= 0 ? vec3(VN[3 * ix.vn], VN[3 * ix.vn + 1], VN[3 * ix.vn + 2]) : vec3(0, 0, 0);
                uv[k] = ix.vt >= 0 ? vec2(VT[2 * ix.vt], VT[2 * ix.vt + 1]) : vec2(0, 0);
            }
            auto same = [](const vec2& a, const vec2& b) { return a.x == b.x && a.y == b.y; };
            if (same(uv[0], uv[1]) || same(uv[1], uv[2]) || same(uv[0], uv[2])) { // Model.cpp:170-175
                uv[0] = vec2(0, 0);
                uv[1] = vec2(1, 0);
                uv[2] = vec2(1, 1);
            }
            tris.push_back(std::make_shared<Triangle>(vs, ns, uv, material));
        }
        meshes.push_back(std::make_shared<Mesh>(sh.name, tris, material));
    }
}

} // namespace Pooraytracer

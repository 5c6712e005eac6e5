// host_api.cpp — host side of the drop-in boundary: the reference's Camera / Hittable / Material
// C++ surface (include/pooraytracer/*.h) implemented on top of the C ABI of libprt_hip.so.
// Nothing here intersects or shades: scene description, flattening, one-time set-up and file output.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "pooraytracer/BVH.h"
#include "pooraytracer/Camera.h"
#include "pooraytracer/Material.h"
#include "pooraytracer/Ray.h"
#include "pooraytracer/SceneFlattener.h"
#include "pooraytracer/Triangle.h"

namespace Pooraytracer {

namespace {
inline double dot3(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross3(const vec3& a, const vec3& b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline vec3 unit3(const vec3& a) { return a * (1.0 / std::sqrt(dot3(a, a))); }
inline bool nan3(const vec3& a) { return a.x != a.x || a.y != a.y || a.z != a.z; }
void check(int rc, const char* what) {
    if (rc != PRT_OK) throw std::runtime_error(std::string(what) + ": " + prt_last_error());
}
} // namespace

// ------------------------------------------------------------------ HitRecord / Hittable
void HitRecord::SetFaceNormal(const Ray& ray, const vec3& outwordNormal) { // Hittable.cpp:8-13
    bFrontFace = dot3(ray.direction, outwordNormal) < 0.;
    normal = bFrontFace ? outwordNormal : -outwordNormal;
}

struct Hittable::DeviceCache {
    SceneFlattener flat;
    PrtScene* scene = nullptr;
    int device = -1;
    unsigned long long sampleCounter = 0;
    ~DeviceCache() {
        if (scene) prt_scene_destroy(scene);
        for (Replica& r : replicas)
            if (r.scene) prt_scene_destroy(r.scene);
    }
    // Additional replicas of the scene for Camera::devices (one PrtScene per extra GPU).
    struct Replica {
        PrtScene* scene = nullptr;
        int device = -1;
    };
    std::vector<Replica> replicas;
    // The `lights` list the scene was created with, as mesh indices (explicitLights false = main.cpp's default list).
    bool explicitLights = false;
    std::vector<int32_t> lightMeshes;
    bool flattened = false;
    void FlattenOnce(const Hittable& h) {
        if (flattened) return;
        h.Flatten(flat);
        flat.EndMesh();
        flattened = true;
    }
    void DescribeWithLights(PrtSceneDesc& d, unsigned flags) {
        flat.Describe(d);
        d.flags = flags;
        static const int32_t none = 0;
        d.light_meshes = explicitLights ? (lightMeshes.empty() ? &none : lightMeshes.data()) : nullptr;
        d.n_light_meshes = explicitLights ? (uint32_t)lightMeshes.size() : 0u;
    }
    // A different lights list needs a different light tree: the scenes are re-created (the traversal BVH with them).
    void SetLights(bool isExplicit, const std::vector<int32_t>& list) {
        if (isExplicit == explicitLights && list == lightMeshes) return;
        explicitLights = isExplicit;
        lightMeshes = list;
        if (scene) prt_scene_destroy(scene);
        scene = nullptr;
        device = -1;
        for (Replica& r : replicas)
            if (r.scene) prt_scene_destroy(r.scene);
        replicas.clear();
    }
    void EnsureReplicas(const std::vector<int>& devs, unsigned flags) { // devs[0] is served by `scene`
        while (replicas.size() + 1 > devs.size()) {
            if (replicas.back().scene) prt_scene_destroy(replicas.back().scene);
            replicas.pop_back();
        }
        replicas.resize(devs.size() - 1);
        for (size_t i = 0; i < replicas.size(); ++i) {
            Replica& r = replicas[i];
            if (!r.scene) {
                PrtSceneDesc d;
                DescribeWithLights(d, flags);
                check(prt_scene_create(&d, &r.scene), "prt_scene_create");
            }
            if (r.device != devs[i + 1]) {
                check(prt_scene_upload(r.scene, devs[i + 1]), "prt_scene_upload");
                r.device = devs[i + 1];
            }
        }
    }
    void Ensure(const Hittable& h, int dev, unsigned flags = 0) {
        if (!scene) {
            FlattenOnce(h);
            PrtSceneDesc d;
            DescribeWithLights(d, flags);
            check(prt_scene_create(&d, &scene), "prt_scene_create");
        }
        if (device != dev) {
            check(prt_scene_upload(scene, dev), "prt_scene_upload");
            device = dev;
        }
    }
};

Hittable::~Hittable() = default;

Hittable::DeviceCache& Hittable::Device() const {
    if (!cache_) cache_ = std::make_shared<DeviceCache>();
    return *cache_;
}

bool Hittable::Hit(const Ray& ray, Interval domain, HitRecord& record) const {
    DeviceCache& dc = Device();
    dc.Ensure(*this, dc.device < 0 ? 0 : dc.device);
    PrtRay r;
    r.o[0] = ray.origin.x; r.o[1] = ray.origin.y; r.o[2] = ray.origin.z;
    r.d[0] = ray.direction.x; r.d[1] = ray.direction.y; r.d[2] = ray.direction.z;
    r.tmin = domain.min;
    r.tmax = domain.max;
    PrtHit h;
    check(prt_trace_closest(dc.scene, &r, 1, &h, 0), "prt_trace_closest");
    if (h.prim < 0) return false;
    const Triangle& t = *dc.flat.triangles[(size_t)h.prim];
    record.position = ray(h.t);
    record.time = h.t;
    record.material = t.material;
    record.tangent = t.tangent;
    record.uv = vec2((1. - h.alpha - h.beta) * t.texCoords[0].x + h.alpha * t.texCoords[1].x + h.beta * t.texCoords[2].x,
                     (1. - h.alpha - h.beta) * t.texCoords[0].y + h.alpha * t.texCoords[1].y + h.beta * t.texCoords[2].y);
    record.SetFaceNormal(ray, t.normal);
    return true;
}

void Hittable::Sample(const point3& origin, HitRecord& rec, double& pdf) const {
    DeviceCache& dc = Device();
    dc.Ensure(*this, dc.device < 0 ? 0 : dc.device);
    const double o[3] = {origin.x, origin.y, origin.z};
    PrtLightSample s;
    check(prt_sample_lights(dc.scene, o, 1, 0x5eed0000ULL + dc.sampleCounter++, &s), "prt_sample_lights");
    rec.position = vec3(s.position[0], s.position[1], s.position[2]);
    rec.normal = vec3(s.normal[0], s.normal[1], s.normal[2]);
    rec.bFrontFace = s.front != 0;
    rec.material = dc.flat.triangles[(size_t)s.prim]->material;
    pdf = s.pdf;
}

// ------------------------------------------------------------------ SceneFlattener
int SceneFlattener::MaterialIndex(const std::shared_ptr<Material>& m) {
    for (size_t i = 0; i < materials.size(); ++i)
        if (materials[i] == m) return (int)i;
    materials.push_back(m);
    return (int)materials.size() - 1;
}
void SceneFlattener::BeginMesh(const std::string& name, const std::shared_ptr<Material>& material) {
    EndMesh();
    meshNames.push_back(name);
    meshMaterial.push_back(MaterialIndex(material));
    inMesh_ = true;
}
void SceneFlattener::EndMesh() {
    if (inMesh_ || looseOpen_) meshFirstTri.push_back(triangles.size());
    inMesh_ = looseOpen_ = false;
}
void SceneFlattener::AddTriangle(const Triangle& t) {
    if (!inMesh_) {
        const int mi = MaterialIndex(t.material);
        if (!looseOpen_ || meshMaterial.back() != mi) {
            EndMesh();
            meshNames.push_back("");
            meshMaterial.push_back(mi);
            looseOpen_ = true;
        }
    }
    for (int k = 0; k < 3; ++k) {
        vertices.insert(vertices.end(), {t.vertices[k].x, t.vertices[k].y, t.vertices[k].z});
        normals.insert(normals.end(), {t.vertexNormals[k].x, t.vertexNormals[k].y, t.vertexNormals[k].z});
        texcoords.insert(texcoords.end(), {t.texCoords[k].x, t.texCoords[k].y});
    }
    triangles.push_back(&t);
}
void SceneFlattener::Describe(PrtSceneDesc& d) {
    EndMesh();
    matTable_.resize(materials.size());
    texTable_.clear();
    textures_.clear();
    for (size_t i = 0; i < materials.size(); ++i) {
        std::shared_ptr<Texture> tex;
        materials[i]->Describe(matTable_[i], &tex);
        matTable_[i].texture = -1;
        if (auto img = std::dynamic_pointer_cast<ImageTexture>(tex)) {
            size_t k = 0;
            for (; k < textures_.size(); ++k)
                if (textures_[k] == tex) break;
            if (k == textures_.size()) {
                textures_.push_back(tex);
                PrtTexture pt;
                pt.width = img->width; pt.height = img->height; pt.channels = img->channels; pt.reserved = 0;
                pt.data = (img->data && !img->data->empty()) ? img->data->data() : nullptr;
                texTable_.push_back(pt);
            }
            matTable_[i].texture = (int32_t)k;
        }
    }
    std::memset(&d, 0, sizeof(d));
    d.n_tris = triangles.size();
    d.vertices = vertices.data();
    d.normals = normals.data();
    d.texcoords = texcoords.data();
    d.n_meshes = (uint32_t)meshMaterial.size();
    d.n_materials = (uint32_t)matTable_.size();
    d.mesh_first_tri = meshFirstTri.data();
    d.mesh_material = meshMaterial.data();
    d.materials = matTable_.data();
    d.n_textures = (uint32_t)texTable_.size();
    d.textures = texTable_.data();
}

// ------------------------------------------------------------------ containers
void HittableList::Flatten(SceneFlattener& out) const {
    for (const auto& o : objects) o->Flatten(out);
}

Triangle::Triangle(const std::array<vec3, 3>& v, const std::array<vec3, 3>& n, const std::array<vec2, 3>& tc,
                   std::shared_ptr<Material> m)
    : vertices(v), texCoords(tc), vertexNormals(n), material(m) { // Triangle.cpp:11-53 (host copies for API users)
    edges[0] = vertices[1] - vertices[0];
    edges[1] = vertices[2] - vertices[0];
    vec3 nn = cross3(edges[0], edges[1]);
    normal = unit3(nn);
    if (nan3(normal)) {
        normal = unit3(n[0] + n[1] + n[2]);
        if (nan3(normal)) normal = vec3(0.0, 0.0, 1.0);
    }
    const double du0 = tc[1].x - tc[0].x, dv0 = tc[1].y - tc[0].y, du1 = tc[2].x - tc[0].x, dv1 = tc[2].y - tc[0].y;
    const double f = 1.0 / (du0 * dv1 - du1 * dv0);
    tangent = unit3(vec3(f * (dv1 * edges[0].x - dv0 * edges[1].x), f * (dv1 * edges[0].y - dv0 * edges[1].y),
                         f * (dv1 * edges[0].z - dv0 * edges[1].z)));
    if (nan3(tangent)) {
        vec3 helper = (std::fabs(normal.x) < (double)0.9f) ? vec3(1, 0, 0) : vec3(0, 1, 0);
        tangent = unit3(cross3(normal, helper));
    }
    area = std::sqrt(dot3(nn, nn)) * 0.5;
    bbox = AABB(AABB(vertices[0], vertices[1]), AABB(vertices[0], vertices[2]));
}
void Triangle::Flatten(SceneFlattener& out) const { out.AddTriangle(*this); }

Mesh::Mesh(const std::string& name_, const std::vector<std::shared_ptr<Hittable>>& triangles, shared_ptr<Material> material_)
    : name(name_), material(material_) {
    for (const auto& p : triangles) Add(p);
}
void Mesh::Flatten(SceneFlattener& out) const {
    out.BeginMesh(name, material);
    for (const auto& o : objects) o->Flatten(out);
    out.EndMesh();
}

BVHNode::BVHNode(HittableList list) : objects_(list.objects) { Init(); }
BVHNode::BVHNode(shared_ptr<Mesh> mesh) : mesh_(mesh), objects_(mesh->objects) { Init(); }
BVHNode::BVHNode(std::vector<shared_ptr<Hittable>>& objects, size_t start, size_t end)
    : objects_(objects.begin() + start, objects.begin() + end) { Init(); }
void BVHNode::Init() {
    bbox = AABB::empty;
    area = 0.0;
    for (const auto& o : objects_) {
        bbox = AABB(bbox, o->BoundingBox());
        area += o->GetArea();
    }
}
void BVHNode::Flatten(SceneFlattener& out) const {
    if (mesh_) mesh_->Flatten(out);
    else
        for (const auto& o : objects_) o->Flatten(out);
}

// ------------------------------------------------------------------ materials
void Material::Zero(PrtMaterial& m, int type) {
    std::memset(&m, 0, sizeof(m));
    m.type = type;
    m.texture = -1;
    m.eta[0] = m.eta[1] = m.eta[2] = 1.0;
    m.alpha_x = m.alpha_y = 0.3;
}
void Lambertian::Describe(PrtMaterial& o, shared_ptr<Texture>* tex) const {
    Zero(o, PRT_MAT_LAMBERTIAN);
    if (texture->IsImage()) *tex = texture; else Put(o.kd, Solid(texture));
}
void DiffuseLight::Describe(PrtMaterial& o, shared_ptr<Texture>*) const {
    Zero(o, PRT_MAT_DIFFUSE_LIGHT);
    Put(o.emission, GetEmission());
}
void PhoneReflectance::Describe(PrtMaterial& o, shared_ptr<Texture>* tex) const {
    Zero(o, PRT_MAT_PHONG);
    o.ns = Ns;
    if (Kd->IsImage()) *tex = Kd;
    else {
        Put(o.kd, Solid(Kd));
        Put(o.ks, Solid(Ks));
    }
}
void PerfectMirror::Describe(PrtMaterial& o, shared_ptr<Texture>*) const { Zero(o, PRT_MAT_MIRROR); }
void CookTorrance::Describe(PrtMaterial& o, shared_ptr<Texture>*) const {
    Zero(o, PRT_MAT_COOKTORRANCE);
    Put(o.kd, Solid(texture));
    Put(o.eta, eta);
    Put(o.k, k);
    o.alpha_x = alphaX;
    o.alpha_y = alphaY;
}
void DebugMaterial::Describe(PrtMaterial& o, shared_ptr<Texture>*) const {
    Zero(o, PRT_MAT_DEBUG);
    Put(o.kd, GetEmission());
}
void EmptyMaterial::Describe(PrtMaterial& o, shared_ptr<Texture>*) const { Zero(o, PRT_MAT_EMPTY); }

// ------------------------------------------------------------------ Camera
void Camera::Render(Hittable& world, Hittable& lights) {
    imageWidth = (imageWidth < 1) ? 1 : imageWidth; // Camera.cpp:77-78
    imageHeight = (imageHeight < 1) ? 1 : imageHeight;
    Hittable::DeviceCache& dc = world.Device();
    const unsigned sceneFlags = bBuildBvhOnDevice ? PRT_SCENE_DEVICE_BVH : 0u;
    const std::vector<int> devs = devices.empty() ? std::vector<int>{device} : devices;
    // `lights` is honoured as given (Camera.cpp:137-139 samples whatever list the caller passes): it is expressed as
    // the list of world meshes it is made of, in its own order.  The list main.cpp:36-45 builds — every emissive mesh,
    // in mesh order — is the library's default and needs nothing extra.  A lights list holding geometry that is not a
    // whole mesh of `world` cannot be expressed through the scene description and is refused, not silently replaced.
    dc.FlattenOnce(world);
    {
        SceneFlattener lf;
        lights.Flatten(lf);
        lf.EndMesh();
        std::map<const Triangle*, int32_t> meshOf;
        for (size_t m = 0; m + 1 < dc.flat.meshFirstTri.size(); ++m)
            for (uint64_t t = dc.flat.meshFirstTri[m]; t < dc.flat.meshFirstTri[m + 1]; ++t) meshOf[dc.flat.triangles[t]] = (int32_t)m;
        std::vector<int32_t> list, dflt;
        for (size_t j = 0; j + 1 < lf.meshFirstTri.size(); ++j) {
            const uint64_t a = lf.meshFirstTri[j], b = lf.meshFirstTri[j + 1];
            if (a == b) continue;
            const auto it = meshOf.find(lf.triangles[a]);
            const int32_t m = it == meshOf.end() ? -1 : it->second;
            bool whole = m >= 0 && (b - a) == (dc.flat.meshFirstTri[m + 1] - dc.flat.meshFirstTri[m]);
            for (uint64_t t = a; t < b && whole; ++t) whole = dc.flat.triangles[dc.flat.meshFirstTri[m] + (t - a)] == lf.triangles[t];
            if (!whole) throw std::invalid_argument("Camera::Render: `lights` must be made of whole meshes of `world` (mesh '" + lf.meshNames[j] + "' is not)");
            list.push_back(m);
        }
        for (size_t m = 0; m < dc.flat.meshMaterial.size(); ++m)
            if (dc.flat.materials[dc.flat.meshMaterial[m]]->HasEmission() && dc.flat.meshFirstTri[m + 1] > dc.flat.meshFirstTri[m]) dflt.push_back((int32_t)m);
        if (list == dflt) dc.SetLights(false, {});
        else dc.SetLights(true, list);
    }
    dc.Ensure(world, devs[0], sceneFlags);
    dc.EnsureReplicas(devs, sceneFlags);
    PrtCamera c;
    c.width = imageWidth; c.height = imageHeight; c.fovy = fovy;
    c.eye[0] = eye.x; c.eye[1] = eye.y; c.eye[2] = eye.z;
    c.look_at[0] = lookAt.x; c.look_at[1] = lookAt.y; c.look_at[2] = lookAt.z;
    c.up[0] = up.x; c.up[1] = up.y; c.up[2] = up.z;
    PrtRenderParams p;
    std::memset(&p, 0, sizeof(p));
    p.spp = samplesPerPixel;
    p.max_depth = maxDepth;
    p.russian_roulette = russianRoulette;
    p.sample_lights = bSampleLights ? 1 : 0;
    p.precision = bFloatPrecision ? PRT_PRECISION_F32 : PRT_PRECISION_F64;
    p.background[0] = background.x; p.background[1] = background.y; p.background[2] = background.z;
    p.seed = seed;
    p.tile_size = 32;
    p.rank = 0;
    p.nranks = 1;
    p.pixel_jitter = bPixelJitter ? 1 : 0;
    std::vector<double> rgb((size_t)imageWidth * imageHeight * 3);
    if (devs.size() == 1) {
        check(prt_render(dc.scene, &c, &p, rgb.data(), nullptr), "prt_render");
    } else {
        // The reference's parallel split is Camera::Render's thread fan-out over row bands (Camera.cpp:46-71); here 16x16
        // tiles are dealt over the devices exactly as bench.py deals them over ranks, every device renders its tiles into
        // a zeroed fp32 framebuffer, and prt_render_multi assembles the frame with ONE RCCL reduce(sum) to the first
        // device (disjoint tiles: x + 0 + ... + 0) and one copy to the host.  The multi-GPU frame is therefore the fp32
        // framebuffer (what travels over xGMI); the per-sample RNG is keyed on the global pixel, so it equals the
        // single-GPU image rounded to float bit for bit.  No host-side sum exists: a failing reduce fails the render.
        std::vector<PrtScene*> scs;
        scs.push_back(dc.scene);
        for (size_t r = 1; r < devs.size(); ++r) scs.push_back(dc.replicas[r - 1].scene);
        std::vector<float> rgb32(rgb.size());
        check(prt_render_multi(scs.data(), (int)scs.size(), &c, &p, rgb32.data()), "prt_render_multi");
        for (size_t i = 0; i < rgb.size(); ++i) rgb[i] = (double)rgb32[i];
    }
    colorAttachment.assign((size_t)imageWidth * imageHeight, color(0., 0., 0.)); // cleared every frame
    for (size_t i = 0; i < colorAttachment.size(); ++i) colorAttachment[i] = color(rgb[i * 3], rgb[i * 3 + 1], rgb[i * 3 + 2]);
    PrtCounters cnt;
    lastRays = 0;
    lastKernelMs = 0.0;
    for (size_t r = 0; r < devs.size(); ++r)
        if (prt_get_counters(r == 0 ? dc.scene : dc.replicas[r - 1].scene, &cnt) == PRT_OK) {
            lastRays += cnt.rays_closest + cnt.rays_shadow;
            lastKernelMs = std::max(lastKernelMs, cnt.kernel_ms); // the devices run concurrently
        }
}

std::string Camera::GetParametersStr() const { // Camera.cpp:332-337
    std::stringstream ss;
    ss << "spp" << samplesPerPixel << "-depth" << maxDepth;
    return ss.str();
}

namespace {
// ---- minimal PNG (stored deflate blocks) and Radiance HDR (flat RGBE) writers; the reference uses stb_image_write
uint32_t crc32(const unsigned char* d, size_t n, uint32_t c = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t k = i;
            for (int j = 0; j < 8; ++j) k = (k & 1) ? 0xEDB88320u ^ (k >> 1) : (k >> 1);
            table[i] = k;
        }
        init = true;
    }
    c = ~c;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ d[i]) & 0xFF] ^ (c >> 8);
    return ~c;
}
void put32(std::vector<unsigned char>& v, uint32_t x) {
    for (int s = 24; s >= 0; s -= 8) v.push_back((unsigned char)(x >> s));
}
void chunk(std::ofstream& f, const char* tag, const std::vector<unsigned char>& data) {
    std::vector<unsigned char> b;
    put32(b, (uint32_t)data.size());
    std::vector<unsigned char> body(tag, tag + 4);
    body.insert(body.end(), data.begin(), data.end());
    b.insert(b.end(), body.begin(), body.end());
    put32(b, crc32(body.data(), body.size()));
    f.write(reinterpret_cast<const char*>(b.data()), (std::streamsize)b.size());
}
bool write_png_rgb8(const std::string& path, int w, int h, const std::vector<uint8_t>& rgb) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    f.write(reinterpret_cast<const char*>(sig), 8);
    std::vector<unsigned char> ihdr;
    put32(ihdr, (uint32_t)w);
    put32(ihdr, (uint32_t)h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    chunk(f, "IHDR", ihdr);
    std::vector<unsigned char> raw;
    raw.reserve((size_t)h * (w * 3 + 1));
    for (int y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb.begin() + (size_t)y * w * 3, rgb.begin() + (size_t)(y + 1) * w * 3);
    }
    std::vector<unsigned char> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (unsigned char c : raw) {
        a = (a + c) % 65521;
        b = (b + a) % 65521;
    }
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xFF));
        z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF));
        z.push_back((unsigned char)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    }
    put32(z, (b << 16) | a);
    chunk(f, "IDAT", z);
    chunk(f, "IEND", {});
    return (bool)f;
}
// Radiance .hdr as the reference's files are: the reference calls stbi_write_hdr (Camera.cpp:327; nothings/stb
// stb_image_write.h, an un-vendored submodule — ThirdPartyLibraries/stb is empty, version unpinned).  Its published format is
// restated here so that a frame written by this library is the same FILE: header with the "Written by" comment and the
// EXPOSURE line, one "new RLE" scanline (2 2 hi lo) per row, the four RGBE components run-length coded one after the
// other (runs of >= 3 equal bytes as (128 + n, byte), n <= 127; everything else as literal dumps of <= 128 bytes), flat
// RGBE pixels for widths outside [8, 32768).  Pinned by the reference's own Results/*.hdr (tests/test_results_pairs.py:
// the decoded floats of a reference file re-encode to the same bytes).
void rgbe_of(const float* c, unsigned char* o) {
    const float m = std::max(c[0], std::max(c[1], c[2]));
    if (m < 1e-32f) {
        o[0] = o[1] = o[2] = o[3] = 0;
    } else {
        int e;
        const float s = (float)std::frexp(m, &e) * 256.0f / m;
        o[0] = (unsigned char)(c[0] * s);
        o[1] = (unsigned char)(c[1] * s);
        o[2] = (unsigned char)(c[2] * s);
        o[3] = (unsigned char)(e + 128);
    }
}
void hdr_scanline(std::vector<unsigned char>& out, int w, const float* rgb, std::vector<unsigned char>& scratch) {
    unsigned char px[4];
    if (w < 8 || w >= 32768) {
        for (int x = 0; x < w; ++x) {
            rgbe_of(rgb + 3 * x, px);
            out.insert(out.end(), px, px + 4);
        }
        return;
    }
    scratch.resize((size_t)w * 4);
    for (int x = 0; x < w; ++x) {
        rgbe_of(rgb + 3 * x, px);
        for (int c = 0; c < 4; ++c) scratch[(size_t)c * w + x] = px[c];
    }
    const unsigned char head[4] = {2, 2, (unsigned char)((w >> 8) & 0xFF), (unsigned char)(w & 0xFF)};
    out.insert(out.end(), head, head + 4);
    for (int c = 0; c < 4; ++c) {
        const unsigned char* comp = &scratch[(size_t)c * w];
        int x = 0;
        while (x < w) {
            int r = x; // start of the next run of three equal bytes (or the end of the row)
            while (r + 2 < w && !(comp[r] == comp[r + 1] && comp[r] == comp[r + 2])) ++r;
            const bool run = r + 2 < w;
            if (!run) r = w;
            while (x < r) { // literals up to the run
                const int n = std::min(128, r - x);
                out.push_back((unsigned char)n);
                out.insert(out.end(), comp + x, comp + x + n);
                x += n;
            }
            if (run) {
                while (r < w && comp[r] == comp[x]) ++r;
                while (x < r) {
                    const int n = std::min(127, r - x);
                    out.push_back((unsigned char)(n + 128));
                    out.push_back(comp[x]);
                    x += n;
                }
            }
        }
    }
}
bool write_hdr(const std::string& path, int w, int h, const std::vector<float>& rgb) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f << "#?RADIANCE\n# Written by stb_image_write.h\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=          1.0000000000000\n\n-Y " << h << " +X " << w << "\n";
    std::vector<unsigned char> out, scratch;
    out.reserve((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) hdr_scanline(out, w, &rgb[(size_t)y * w * 3], scratch);
    f.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)out.size());
    return (bool)f;
}
double linear_to_srgb(double c) { // Camera.cpp:214-221
    if (c <= 0.0031308) return 12.92 * c;
    return 1.055 * std::pow(c, (1. / 2.4)) - 0.055;
}
} // namespace

void Camera::WriteColorAttachment(const std::string& outputPath, bool bWriteHDR) const { // Camera.cpp:279-331
    std::vector<uint8_t> raw((size_t)imageWidth * imageHeight * 3);
    std::vector<float> hdr(bWriteHDR ? raw.size() : 0);
    const Interval intensity(0.0000, 0.9999);
    for (size_t i = 0; i < (size_t)imageWidth * imageHeight && i < colorAttachment.size(); ++i) {
        double c[3] = {colorAttachment[i].x, colorAttachment[i].y, colorAttachment[i].z};
        for (int k = 0; k < 3; ++k) {
            if (c[k] != c[k]) c[k] = 0.0;
            raw[i * 3 + k] = (uint8_t)(intensity.Clamp(linear_to_srgb(c[k])) * 255);
            if (bWriteHDR) hdr[i * 3 + k] = (float)c[k];
        }
    }
    if (!write_png_rgb8(outputPath, imageWidth, imageHeight, raw)) std::fprintf(stderr, "[pooraytracer] cannot write %s\n", outputPath.c_str());
    if (bWriteHDR) {
        const std::string h = outputPath.substr(0, outputPath.find_last_of('.')) + ".hdr";
        if (!write_hdr(h, imageWidth, imageHeight, hdr)) std::fprintf(stderr, "[pooraytracer] cannot write %s\n", h.c_str());
    }
}

namespace {
// attribute="value" lookup inside the first <tag ...> element at or after `from`
bool find_tag(const std::string& s, const std::string& tag, size_t from, size_t& b, size_t& e) {
    b = s.find("<" + tag, from);
    if (b == std::string::npos) return false;
    e = s.find('>', b);
    return e != std::string::npos;
}
bool attr(const std::string& s, size_t b, size_t e, const std::string& name, double& out) {
    size_t p = b;
    while ((p = s.find(name, p)) != std::string::npos && p < e) {
        const bool word = (p == 0 || std::isspace((unsigned char)s[p - 1]));
        size_t q = p + name.size();
        while (q < e && std::isspace((unsigned char)s[q])) ++q;
        if (word && q < e && s[q] == '=') {
            q = s.find_first_of("\"'", q);
            if (q == std::string::npos || q > e) return false;
            out = std::strtod(s.c_str() + q + 1, nullptr);
            return true;
        }
        p += name.size();
    }
    return false;
}
} // namespace

void Camera::SetViewParametersByXmlFile(const std::string& xmlFilePath) { // Camera.cpp:339-389
    std::ifstream f(xmlFilePath);
    if (!f) {
        std::fprintf(stderr, "[pooraytracer] Failed to load XML file: %s\n", xmlFilePath.c_str());
        return;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string s = ss.str();
    size_t b, e;
    if (!find_tag(s, "camera", 0, b, e)) return;
    double v;
    if (attr(s, b, e, "width", v)) imageWidth = (int)v;
    if (attr(s, b, e, "height", v)) imageHeight = (int)v;
    if (attr(s, b, e, "fovy", v)) fovy = v;
    auto vec = [&](const char* tag, vec3& dst) {
        size_t tb, te;
        if (!find_tag(s, tag, e, tb, te)) return;
        double x = dst.x, y = dst.y, z = dst.z;
        attr(s, tb, te, "x", x);
        attr(s, tb, te, "y", y);
        attr(s, tb, te, "z", z);
        dst = vec3(x, y, z);
    };
    vec("eye", eye);
    vec("lookat", lookAt);
    vec("up", up);
}

} // namespace Pooraytracer

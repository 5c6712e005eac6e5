// render_scene.cpp — the reference's main.cpp flow (main.cpp:24-52) written against the drop-in host
// API: build Triangle / Mesh / Material objects, wrap every mesh in a BVHNode, put emissive meshes in
// `lights`, wrap both lists in a top-level BVHNode, camera.Render(world, lights), write PNG + HDR.
//
// The reference loads <scene>.obj/.mtl/.xml with tinyobjloader / tinyxml2 (Source/Model.cpp); that
// loader is a "next" row, so this driver reads a flat binary scene dump instead
// (written by pooraytracer_amd.scenes.dump_scene) and an optional camera XML in the reference's format.
//
// usage: render_scene <scene.bin> <spp> <depth> <out.f64> [out.png] [camera.xml]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "pooraytracer/BVH.h"
#include "pooraytracer/Camera.h"
#include "pooraytracer/Material.h"
#include "pooraytracer/Ray.h"
#include "pooraytracer/Triangle.h"

using namespace Pooraytracer;

template <typename T>
static T rd(std::ifstream& f) {
    T v;
    f.read(reinterpret_cast<char*>(&v), sizeof(T));
    if (!f) throw std::runtime_error("scene dump truncated");
    return v;
}

int main(int argc, char** argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s scene.bin spp depth out.f64 [out.png] [camera.xml]\n", argv[0]);
        return 2;
    }
    try {
        std::ifstream f(argv[1], std::ios::binary);
        if (!f) throw std::runtime_error("cannot open scene dump");
        if (rd<uint32_t>(f) != 0x50525431u) throw std::runtime_error("bad magic");
        Camera camera;
        camera.bSampleLights = true; // main.cpp:24-30
        camera.russianRoulette = 0.8;
        camera.samplesPerPixel = std::atoi(argv[2]);
        camera.maxDepth = std::atoi(argv[3]);
        camera.threadNums = 16;
        camera.background = color(0.0, 0.0, 0.0);
        camera.imageWidth = rd<int32_t>(f);
        camera.imageHeight = rd<int32_t>(f);
        camera.fovy = rd<double>(f);
        double v[9];
        for (double& x : v) x = rd<double>(f);
        camera.eye = vec3(v[0], v[1], v[2]);
        camera.lookAt = vec3(v[3], v[4], v[5]);
        camera.up = vec3(v[6], v[7], v[8]);
        if (argc > 6) camera.SetViewParametersByXmlFile(argv[6]);
        // additions of this build, driven from the environment so the argument list stays main.cpp-like
        if (const char* e = std::getenv("PRT_EXAMPLE_F32")) camera.bFloatPrecision = std::atoi(e) != 0; // fp32 fast mode
        if (const char* e = std::getenv("PRT_EXAMPLE_DEVICES")) { // e.g. "0,1,2,3": tiles over several GPUs
            for (const char* q = e; *q;) {
                camera.devices.push_back(std::atoi(q));
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
        }
        if (const char* e = std::getenv("PRT_EXAMPLE_DEVICE_BVH")) camera.bBuildBvhOnDevice = std::atoi(e) != 0;

        const uint32_t nTex = rd<uint32_t>(f);
        std::vector<std::shared_ptr<Texture>> textures;
        for (uint32_t i = 0; i < nTex; ++i) {
            const int w = rd<int32_t>(f), h = rd<int32_t>(f), c = rd<int32_t>(f);
            std::vector<unsigned char> px((size_t)w * h * c);
            f.read(reinterpret_cast<char*>(px.data()), (std::streamsize)px.size());
            textures.push_back(std::make_shared<ImageTexture>(w, h, c, px.data()));
        }
        const uint32_t nMat = rd<uint32_t>(f);
        std::vector<std::shared_ptr<Material>> materials;
        for (uint32_t i = 0; i < nMat; ++i) {
            const int type = rd<int32_t>(f), tex = rd<int32_t>(f);
            double p[18];
            for (double& x : p) x = rd<double>(f);
            const color kd(p[0], p[1], p[2]), ks(p[3], p[4], p[5]), em(p[7], p[8], p[9]);
            const double ns = p[6];
            const vec3 eta(p[10], p[11], p[12]), k(p[13], p[14], p[15]);
            switch (type) { // the material factory of Source/Model.cpp:278-330
            case PRT_MAT_LAMBERTIAN:
                materials.push_back(tex >= 0 ? std::make_shared<Lambertian>(textures[tex]) : std::make_shared<Lambertian>(kd));
                break;
            case PRT_MAT_PHONG:
                materials.push_back(tex >= 0 ? std::make_shared<PhoneReflectance>(textures[tex], ks, ns)
                                             : std::make_shared<PhoneReflectance>(kd, ks, ns));
                break;
            case PRT_MAT_MIRROR: materials.push_back(std::make_shared<PerfectMirror>()); break;
            case PRT_MAT_COOKTORRANCE: materials.push_back(std::make_shared<CookTorrance>(kd, p[16], p[17], eta, k)); break;
            case PRT_MAT_DIFFUSE_LIGHT: materials.push_back(std::make_shared<DiffuseLight>(em)); break;
            case PRT_MAT_DEBUG: materials.push_back(std::make_shared<DebugMaterial>(kd)); break;
            default: materials.push_back(std::make_shared<EmptyMaterial>()); break;
            }
        }
        const uint32_t nMesh = rd<uint32_t>(f);
        std::vector<std::shared_ptr<Mesh>> meshes;
        for (uint32_t m = 0; m < nMesh; ++m) {
            const uint32_t nameLen = rd<uint32_t>(f);
            std::string name(nameLen, ' ');
            f.read(name.data(), nameLen);
            const int mat = rd<int32_t>(f);
            const uint64_t nTri = rd<uint64_t>(f);
            std::vector<std::shared_ptr<Hittable>> tris;
            for (uint64_t t = 0; t < nTri; ++t) {
                double d[24];
                for (double& x : d) x = rd<double>(f);
                std::array<vec3, 3> vs{vec3(d[0], d[1], d[2]), vec3(d[3], d[4], d[5]), vec3(d[6], d[7], d[8])};
                std::array<vec3, 3> ns{vec3(d[9], d[10], d[11]), vec3(d[12], d[13], d[14]), vec3(d[15], d[16], d[17])};
                std::array<vec2, 3> uv{vec2(d[18], d[19]), vec2(d[20], d[21]), vec2(d[22], d[23])};
                tris.push_back(std::make_shared<Triangle>(vs, ns, uv, materials[mat]));
            }
            meshes.push_back(std::make_shared<Mesh>(name, tris, materials[mat]));
        }

        // main.cpp:35-45
        HittableList world;
        HittableList lights;
        for (auto& mesh : meshes) {
            world.Add(make_shared<BVHNode>(mesh));
            if (mesh->material->HasEmission()) lights.Add(make_shared<BVHNode>(mesh));
        }
        world = HittableList(make_shared<BVHNode>(world));
        lights = HittableList(make_shared<BVHNode>(lights));

        camera.Render(world, lights); // main.cpp:49
        std::ofstream o(argv[4], std::ios::binary);
        o.write(reinterpret_cast<const char*>(camera.colorAttachment.data()),
                (std::streamsize)(camera.colorAttachment.size() * sizeof(color)));
        if (argc > 5) camera.WriteColorAttachment(argv[5]); // main.cpp:52
        std::printf("rendered %dx%d %s: %llu rays, kernel %.3f ms\n", camera.imageWidth, camera.imageHeight,
                    camera.GetParametersStr().c_str(), camera.lastRays, camera.lastKernelMs);
        // single-ray queries through the same boundary (world.Hit / lights.Sample)
        HitRecord rec;
        Ray centre(camera.eye, camera.lookAt - camera.eye);
        if (world.Hit(centre, Interval(0.0001, 1e30), rec)) std::printf("centre ray hits at t=%.17g frontFace=%d\n", rec.time, (int)rec.bFrontFace);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}

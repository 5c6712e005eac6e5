// Test helper: main.cpp's flow, but with a `lights` list chosen by the caller instead of "every emissive mesh":
//   lights_arg <resources_dir> <scene> <mode> <spp> <depth> <out.f64>
// mode: "all" (main.cpp:36-45), "first" (only the first emissive mesh), "reversed" (emissive meshes in reverse order),
//       "foreign" (a mesh that is not part of world: must be refused).
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <string>

#include "pooraytracer/BVH.h"
#include "pooraytracer/Camera.h"
#include "pooraytracer/Model.h"

int main(int argc, char** argv) {
    using namespace Pooraytracer;
    if (argc < 7) return 2;
    try {
        const std::string name = argv[2], path = std::string(argv[1]) + "/" + name, mode = argv[3];
        Camera camera;
        camera.bSampleLights = true;
        camera.russianRoulette = 0.8;
        camera.samplesPerPixel = std::atoi(argv[4]);
        camera.maxDepth = std::atoi(argv[5]);
        camera.SetViewParametersByXmlFile(path + "/" + name + ".xml");
        auto model = std::make_shared<Model>(path, name);
        HittableList world, lights;
        std::vector<std::shared_ptr<Mesh>> emissive;
        for (auto& mesh : model->meshes) {
            world.Add(make_shared<BVHNode>(mesh));
            if (mesh->material->HasEmission()) emissive.push_back(mesh);
        }
        if (mode == "first") emissive.resize(std::min<size_t>(1, emissive.size()));
        if (mode == "reversed") std::reverse(emissive.begin(), emissive.end());
        for (auto& mesh : emissive) lights.Add(make_shared<BVHNode>(mesh));
        if (mode == "foreign") {
            auto other = std::make_shared<Model>(path, name); // same file, different objects
            lights.Add(make_shared<BVHNode>(other->meshes.back()));
        }
        world = HittableList(make_shared<BVHNode>(world));
        lights = HittableList(make_shared<BVHNode>(lights));
        camera.Render(world, lights);
        std::ofstream o(argv[6], std::ios::binary);
        o.write(reinterpret_cast<const char*>(camera.colorAttachment.data()), (std::streamsize)(camera.colorAttachment.size() * sizeof(color)));
        std::printf("ok %s\n", mode.c_str());
        return 0;
    } catch (const std::exception& e) {
        std::printf("refused: %s\n", e.what());
        return 1;
    }
}

// pooraytracer_main.cpp — the reference's main.cpp (main.cpp:6-55) against the drop-in host API, with
// the hard-coded scene name / spp / depth turned into arguments:
//   pooraytracer_main <resources_dir> <scene_name> [spp=100] [depth=100] [out_dir=.] [out.f64]
// Reads <resources_dir>/<scene>/<scene>.obj|.mtl|.xml like the reference, renders on the GPU, writes
// <scene>_spp<S>-depth<D>_<seconds>s.png + .hdr (main.cpp:52 naming, timestamp omitted).
#include <chrono>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>

#include "pooraytracer/BVH.h"
#include "pooraytracer/Camera.h"
#include "prt.h"
#include "pooraytracer/Model.h"

int main(int argc, char** argv) {
    using namespace Pooraytracer;
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s resources_dir scene_name [spp] [depth] [out_dir] [out.f64]\n", argv[0]);
        return 2;
    }
    try {
        const std::string fileName = argv[2];
        const std::string filePath = std::string(argv[1]) + "/" + fileName;
        Camera camera;
        camera.bSampleLights = true;
        camera.russianRoulette = 0.8;
        camera.samplesPerPixel = argc > 3 ? std::atoi(argv[3]) : 100;
        camera.maxDepth = argc > 4 ? std::atoi(argv[4]) : 100;
        camera.threadNums = 16;
        camera.background = color(0.0, 0.0, 0.0);
        camera.SetViewParametersByXmlFile(filePath + "/" + fileName + ".xml");

        std::shared_ptr<Model> model = std::make_shared<Model>(filePath, fileName);
        HittableList world;
        HittableList lights;
        for (auto& mesh : model->meshes) {
            world.Add(make_shared<BVHNode>(mesh));
            if (mesh->material->HasEmission()) lights.Add(make_shared<BVHNode>(mesh));
        }
        world = HittableList(make_shared<BVHNode>(world));
        lights = HittableList(make_shared<BVHNode>(lights));

        auto start = std::chrono::steady_clock::now();
        camera.Render(world, lights);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
        char t[64];
        std::snprintf(t, sizeof(t), "%.2fs", sec);
        const std::string outDir = argc > 5 ? argv[5] : ".";
        const std::string png = outDir + "/" + fileName + "_" + camera.GetParametersStr() + "_" + t + ".png";
        camera.WriteColorAttachment(png);
        if (argc > 6) {
            std::ofstream o(argv[6], std::ios::binary);
            o.write(reinterpret_cast<const char*>(camera.colorAttachment.data()),
                    (std::streamsize)(camera.colorAttachment.size() * sizeof(color)));
        }
        std::printf("%s: %zu meshes, %dx%d %s, %.3f s (first Render includes BVH build + upload), %llu rays, kernel %.2f ms -> %s\n",
                    fileName.c_str(), model->meshes.size(), camera.imageWidth, camera.imageHeight,
                    camera.GetParametersStr().c_str(), sec, camera.lastRays, camera.lastKernelMs, png.c_str());
        prt_shutdown(); // releases the RCCL communicators a multi-device Camera::devices render cached (no-op otherwise)
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}

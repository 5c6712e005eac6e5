// Test helper: load <dir>/<name>/<name>.{obj,mtl} with the host loader and print every mesh (name, material class
// via its emission / scatter flags) and triangle (9 vertex coordinates, 6 texture coordinates, %.17g).
// Exit 0 = loaded, 1 = the loader threw (message on stdout).
#include <cstdio>
#include <exception>
#include <string>

#include "pooraytracer/Material.h"
#include "pooraytracer/Model.h"
#include "pooraytracer/Triangle.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    try {
        Pooraytracer::Model model(argv[1], argv[2]);
        for (const auto& m : model.meshes) {
            std::printf("mesh %s emission %d skipNEE %d tris %zu\n", m->name.c_str(), m->material->HasEmission() ? 1 : 0,
                        m->material->SkipLightSampling() ? 1 : 0, m->objects.size());
            for (const auto& h : m->objects) {
                const auto* t = dynamic_cast<const Pooraytracer::Triangle*>(h.get());
                if (!t) return 3;
                std::printf("t");
                for (int k = 0; k < 3; ++k) std::printf(" %.17g %.17g %.17g", t->vertices[k].x, t->vertices[k].y, t->vertices[k].z);
                for (int k = 0; k < 3; ++k) std::printf(" %.17g %.17g", t->texCoords[k].x, t->texCoords[k].y);
                std::printf("\n");
            }
        }
        return 0;
    } catch (const std::exception& e) {
        std::printf("refused: %s\n", e.what());
        return 1;
    }
}

// Test helper (built with -fsanitize=address,undefined): load <dir>/<name>/<name>.{obj,mtl,xml} with the host
// loader.  Exit 0 = loaded, 1 = the loader threw (malformed input is allowed to be refused), anything else = bug.
#include <cstdio>
#include <exception>
#include <string>

#include "pooraytracer/Camera.h"
#include "pooraytracer/Model.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    try {
        Pooraytracer::Model model(argv[1], argv[2]);
        size_t tris = 0;
        for (const auto& m : model.meshes) tris += m->objects.size();
        Pooraytracer::Camera cam;
        cam.SetViewParametersByXmlFile(std::string(argv[1]) + "/" + argv[2] + ".xml");
        std::printf("meshes %zu triangles %zu camera %dx%d\n", model.meshes.size(), tris, cam.imageWidth, cam.imageHeight);
        return 0;
    } catch (const std::exception& e) {
        std::printf("refused: %s\n", e.what());
        return 1;
    }
}

// Test helper: Camera::WriteColorAttachment (the product's output stage, Camera.cpp:279-331) on caller-supplied linear floats.
// usage: output_check <w> <h> <rgb_f32.raw> <out.png>   -> writes out.png and out.hdr
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pooraytracer/Camera.h"
int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int w = std::atoi(argv[1]), h = std::atoi(argv[2]);
    std::vector<float> px((size_t)w * h * 3);
    FILE* f = std::fopen(argv[3], "rb");
    if (!f || std::fread(px.data(), sizeof(float), px.size(), f) != px.size()) return 1;
    std::fclose(f);
    Pooraytracer::Camera cam;
    cam.imageWidth = w;
    cam.imageHeight = h;
    cam.colorAttachment.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; ++i) cam.colorAttachment[i] = Pooraytracer::color(px[3 * i], px[3 * i + 1], px[3 * i + 2]);
    cam.WriteColorAttachment(argv[4], true);
    return 0;
}

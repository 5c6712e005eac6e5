// Camera.h — mirror of Source/Camera.h:11-53: same public fields, Render(world, lights) and
// colorAttachment.  Render flattens `world` (cached per object graph), uploads it once, and runs the
// HIP path tracer through the C ABI (prt_render); colorAttachment receives the fp64 framebuffer.
#pragma once
#include <string>
#include <vector>

#include "HittableList.h"
#include "Math.h"

namespace Pooraytracer {
class Camera {
public:
    int imageWidth = 100;
    int imageHeight = 100;
    int samplesPerPixel = 1;
    int threadNums = 16; // kept for source compatibility; the device schedules its own wavefronts
    int maxDepth = 10;
    color background = color(0., 0., 0.);

    double fovy = 90.;
    vec3 eye = vec3(0., 0., 0.);
    vec3 lookAt = vec3(0., 0., -1.);
    vec3 up = vec3(0., 1., 0.);

    std::vector<color> colorAttachment;
    void Render(Hittable& world, Hittable& lights);
    // 8-bit sRGB PNG (+ Radiance .hdr), Camera.cpp:279-331
    void WriteColorAttachment(const std::string& outputPath, bool bWriteHDR = true) const;
    std::string GetParametersStr() const;
    // <camera width height fovy><eye/><lookat/><up/></camera>, Camera.cpp:339-389
    void SetViewParametersByXmlFile(const std::string& xmlFilePath);

    bool bSampleLights = true;
    double russianRoulette = 0.8;

    // additions (not in the reference): RNG key, device index, counters of the last Render
    unsigned long long seed = 1;
    bool bBuildBvhOnDevice = false; // PRT_SCENE_DEVICE_BVH: build the traversal BVH on the GPU (first Render of a world)
    bool bFloatPrecision = false; // PRT_PRECISION_F32: the fp32 fast mode (tolerance tier 2; the reference computes in double)
    bool bPixelJitter = false; // per-sample SampleSquare() pixel offset: the AA the reference has commented out (Camera.cpp:110-111)
    int device = 0;
    std::vector<int> devices; // non-empty: cut the frame into tiles over these GPUs (one host thread each); overrides `device`
    unsigned long long lastRays = 0;
    double lastKernelMs = 0.0;
};
} // namespace Pooraytracer

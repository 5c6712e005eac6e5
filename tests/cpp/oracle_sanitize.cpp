// Test helper: exercises the CPU oracle (built with -fsanitize=address,undefined) on a tiny scene:
// closest hits, light sampling, a multi-threaded render.  Prints a checksum; exit code 0 on success.
#include <cmath>
#include <cstdio>
#include <vector>

#include "oracle.h"

int main() {
    // floor quad (2 triangles, Lambertian), a mirror triangle, a Phong triangle and a light triangle
    const double V[5][3][3] = {{{-2, 0, -2}, {-2, 0, 2}, {2, 0, 2}},     {{-2, 0, -2}, {2, 0, 2}, {2, 0, -2}},
                               {{-1, 0.1, -1}, {0, 1.2, -1.5}, {1, 0.1, -1}}, {{0.2, 0.05, 0.3}, {0.9, 0.6, 0.2}, {0.8, 0.05, 0.9}},
                               {{-0.5, 2, -0.5}, {0.5, 2, -0.5}, {0, 2, 0.5}}};
    const double UV[5][3][2] = {{{0, 0}, {0, 1}, {1, 1}}, {{0, 0}, {1, 1}, {1, 0}}, {{0, 0}, {0.5, 1}, {1, 0}},
                                {{0, 0}, {1, 0}, {1, 1}}, {{0, 0}, {0, 0}, {0, 0}}}; // last: degenerate uv -> tangent fallback
    OrcMaterial mats[4] = {};
    mats[0].type = 0; mats[0].texture = 0; mats[0].kd[0] = mats[0].kd[1] = mats[0].kd[2] = 0.7;
    mats[1].type = 2; mats[1].texture = -1;
    mats[2].type = 1; mats[2].texture = -1; mats[2].kd[0] = 0.5; mats[2].kd[1] = 0.4; mats[2].kd[2] = 0.3;
    mats[2].ks[0] = mats[2].ks[1] = mats[2].ks[2] = 0.4; mats[2].ns = 40;
    mats[3].type = 4; mats[3].texture = -1; mats[3].emission[0] = 12; mats[3].emission[1] = 10; mats[3].emission[2] = 8;
    unsigned char texels[3 * 3 * 3];
    for (int i = 0; i < 27; ++i) texels[i] = (unsigned char)(i * 9);
    OrcTexture tex = {3, 3, 3, 0, texels};
    const uint64_t first[5] = {0, 2, 3, 4, 5};
    const int32_t mm[4] = {0, 1, 2, 3};
    OrcSceneDesc d = {};
    d.n_tris = 5; d.vertices = &V[0][0][0]; d.texcoords = &UV[0][0][0];
    d.n_meshes = 4; d.n_materials = 4; d.mesh_first_tri = first; d.mesh_material = mm; d.materials = mats;
    d.n_textures = 1; d.textures = &tex;
    OrcScene* s = orc_scene_create(&d);
    if (!s || orc_light_count(s) != 1) return 2;
    OrcRay rays[3] = {{{0, 3, 0}, 1e-4, {0, -1, 0}, INFINITY}, {{0, 0.5, 3}, 1e-4, {0, 0, -1}, INFINITY}, {{5, 5, 5}, 1e-4, {1, 0, 0}, INFINITY}};
    OrcHit hits[3];
    orc_trace_closest(s, rays, 3, hits);
    if (hits[0].prim != 4 || hits[2].prim != -1) return 3;
    double org[6] = {0, 0, 0, 0.5, 0.01, 0.5};
    OrcLightSample ls[2];
    orc_sample_lights(s, org, 2, 7, ls);
    OrcCamera cam = {24, 16, 45.0, {0.3, 1.5, 3.5}, {0, 0.4, 0}, {0, 1, 0}};
    OrcRenderParams p = {};
    p.spp = 6; p.max_depth = 8; p.russian_roulette = 0.8; p.sample_lights = 1; p.seed = 3; p.tile_size = 32; p.nranks = 1;
    std::vector<double> img(24 * 16 * 3);
    OrcCounters c;
    orc_render(s, &cam, &p, img.data(), 0, 16, 3, 1, &c);
    p.pixel_jitter = 1;
    orc_render(s, &cam, &p, img.data(), 0, 16, 1, 0, &c);
    double sum = 0;
    for (double x : img) sum += x;
    orc_scene_destroy(s);
    if (!(sum == sum) || sum <= 0) return 4;
    std::printf("ok %.6f rays %llu\n", sum, (unsigned long long)(c.rays_closest + c.rays_shadow));
    return 0;
}

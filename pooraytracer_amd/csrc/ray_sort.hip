// K4 — active-ray sorting for ray batches (SURVEY.md §2 "K4 optional"; north_star: "ballot / compaction for active-ray
// sorting").  K1 hands consecutive rays of the batch to the lanes of a wave; when the scene does not fit the caches
// (S4-class: BVH + triangles beyond L2 and Infinity Cache) and the batch is incoherent, every node a lane visits is a
// line of its own from HBM.  This pre-pass gives K1 a permutation of the batch in which consecutive rays start close
// together and point the same way: key = 27-bit Morton code of the origin's cell in the scene's box (9 bits per axis) +
// the direction's octant (3 bits), (key, index) pairs sorted by rocprim's radix sort — the library primitive the device
// BVH build uses as well.  K1 then reads ray perm[i] and writes hit perm[i]: same rays, same hits, other order.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/prt.h"

namespace prt {

namespace {
__device__ __forceinline__ uint32_t spread9(uint32_t v) { // 9 bits -> every third bit
    v &= 0x1ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ void k_ray_keys(const PrtRay* __restrict__ rays, uint32_t n, float gx, float gy, float gz, float sx, float sy, float sz,
                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double4* rp = reinterpret_cast<const double4*>(rays + i);
    const double4 r0 = rp[0], r1 = rp[1];
    // cell of the origin in a 512^3 grid over the scene's box (origins outside it go to the boundary cells; NaN -> 0)
    // (relative to the grid origin in the precision the origin comes in, like the box test: the cells of a scene far from the
    // world origin are as fine as those of the same scene at the origin)
    const float cx = fminf(fmaxf((float)(r0.x - (double)gx) * sx, 0.f), 511.f);
    const float cy = fminf(fmaxf((float)(r0.y - (double)gy) * sy, 0.f), 511.f);
    const float cz = fminf(fmaxf((float)(r0.z - (double)gz) * sz, 0.f), 511.f);
    const uint32_t m = spread9((uint32_t)cx) | (spread9((uint32_t)cy) << 1) | (spread9((uint32_t)cz) << 2);
    const uint32_t oct = (r1.x < 0.0 ? 1u : 0u) | (r1.y < 0.0 ? 2u : 0u) | (r1.z < 0.0 ? 4u : 0u);
    keys[i] = (m << 3) | oct;
    vals[i] = i;
}
} // namespace

// Bytes of scratch ray_sort needs for n rays (four n-word arrays + rocprim's own).
size_t ray_sort_scratch_bytes(size_t n, std::string* err) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, p, p, p, p, n, 0u, 30u, nullptr);
    if (e != hipSuccess) {
        if (err) *err = std::string("rocprim::radix_sort_pairs (size query): ") + hipGetErrorString(e);
        return 0;
    }
    return 4 * ((n * sizeof(uint32_t) + 255) / 256 * 256) + bytes;
}

// Writes the permutation to the first n words of `scratch` (ray_sort_scratch_bytes(n) bytes, 256-byte aligned); returns it.
const uint32_t* ray_sort(const PrtRay* d_rays, size_t n, const float grid_origin[3], const float grid_step[3], void* scratch,
                         size_t scratch_bytes, hipStream_t st, std::string* err) {
    const size_t words = (n * sizeof(uint32_t) + 255) / 256 * 256;
    unsigned char* base = static_cast<unsigned char*>(scratch);
    uint32_t* order = reinterpret_cast<uint32_t*>(base);
    uint32_t* keys = reinterpret_cast<uint32_t*>(base + words);
    uint32_t* keys2 = reinterpret_cast<uint32_t*>(base + 2 * words);
    uint32_t* vals = reinterpret_cast<uint32_t*>(base + 3 * words);
    void* tmp = base + 4 * words;
    size_t tmp_bytes = scratch_bytes - 4 * words;
    float s[3];
    for (int a = 0; a < 3; ++a) {
        const float extent = grid_step[a] * 65536.f;
        s[a] = extent > 0.f ? 512.f / extent : 0.f;
    }
    hipLaunchKernelGGL(k_ray_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_rays, (uint32_t)n, grid_origin[0], grid_origin[1],
                       grid_origin[2], s[0], s[1], s[2], keys, vals);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys2, vals, order, n, 0u, 30u, st);
    if (e != hipSuccess) {
        if (err) *err = std::string("ray_sort: ") + hipGetErrorString(e);
        return nullptr;
    }
    return order;
}

} // namespace prt

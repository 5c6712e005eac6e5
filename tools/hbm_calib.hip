// Developer tool: calibrates what the memory-side counters (FETCH_SIZE, TCC_MISS, TCC_EA0_RDREQ*) report for the
// access patterns of BVH traversal on a scene that does not fit the caches — scattered 32- / 64- / 128-byte reads —
// against a streaming read of known size, and measures the rate at which MI355X serves each of them.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_calib.hip -o tools/hbm_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- tools/hbm_calib        (one --pmc set per run)
// Every kernel reads `bytes_per_item` bytes per lane: stream = consecutive 16-byte pieces; gatherN = N bytes at a
// hashed N-byte-aligned offset of a 2 GiB buffer (8x the Infinity Cache), one item per lane, 2^26 items per launch.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));      \
            return 1;                                                         \
        }                                                                     \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 27; z *= 0x94D049BB133111EBULL; z ^= z >> 31;
    return z;
}

__global__ void k_stream(const uint4* __restrict__ buf, size_t n16, unsigned long long* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = buf[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ULL);
}

template <int BYTES>
__global__ void k_gather(const uint4* __restrict__ buf, size_t buf_bytes, size_t items, unsigned long long* sink) {
    uint32_t acc = 0;
    const size_t slots = buf_bytes / BYTES;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < items; i += (size_t)gridDim.x * blockDim.x) {
        const size_t slot = mix(i + 1) % slots;
        const uint4* p = buf + slot * (BYTES / 16);
#pragma unroll
        for (int k = 0; k < BYTES / 16; ++k) {
            const uint4 v = p[k];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ULL);
}

int main() {
    const size_t buf_bytes = 2ull << 30;
    const size_t items = 1ull << 26;
    uint4* d = nullptr;
    unsigned long long* sink = nullptr;
    CK(hipMalloc(&d, buf_bytes));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(d, 1, buf_bytes));
    CK(hipMemset(sink, 0, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 16, block = 256;
    auto time = [&](const char* name, double useful_bytes, auto launch) -> int {
        launch();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        std::printf("{\"kernel\": \"%s\", \"ms\": %.4f, \"useful_bytes\": %.0f, \"useful_GBps\": %.1f, \"items_per_us\": %.1f}\n", name, best,
                    useful_bytes, useful_bytes / best / 1e6, name[0] == 's' ? 0.0 : (double)items / best / 1e3);
        return 0;
    };
    if (time("stream_2GiB", (double)buf_bytes, [&] { k_stream<<<grid, block>>>(d, buf_bytes / 16, sink); })) return 1;
    if (time("gather32", 32.0 * items, [&] { k_gather<32><<<grid, block>>>(d, buf_bytes, items, sink); })) return 1;
    if (time("gather64", 64.0 * items, [&] { k_gather<64><<<grid, block>>>(d, buf_bytes, items, sink); })) return 1;
    if (time("gather128", 128.0 * items, [&] { k_gather<128><<<grid, block>>>(d, buf_bytes, items, sink); })) return 1;
    return 0;
}

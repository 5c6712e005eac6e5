// Developer tool (round 4): does a CDNA4 SIMD issue a vector instruction faster when part of the wave's EXEC mask is zero?
// K3's node rounds run at 0.41-0.61 lane utilisation: if empty 16-lane rows were skipped, idle lanes would be
// cheaper than a full instruction slot and packing the active lanes of a wave together would pay; if not, an instruction
// costs the same whether 1 or 64 lanes execute it.
//   hipcc -O3 --offload-arch=gfx950 tools/exec_mask_calib.hip -o tools/exec_mask_calib && tools/exec_mask_calib > profiles/r04_exec_mask_calibration.json
// Streams of independent instructions of one kind (16 accumulators) under different EXEC masks, 3 waves per SIMD (one
// 768-thread workgroup per CU), timed with the shader clock by lane 0 of every wave.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int ITER = 2048;
enum Op : int { FMA_F32, FMA_F64, CVT_F32_U32, ALIGNBIT, N_OPS };
static const char* kNames[N_OPS] = {"v_fma_f32", "v_fma_f64", "v_cvt_f32_u32", "v_alignbit_b32"};

template <int OP>
__global__ __launch_bounds__(1024) void k_masked(unsigned long long* cycles, float* sink, float seed, unsigned long long mask_in, int mixed) {
    const unsigned long long mask = (mixed && (threadIdx.x >> 6) >= 4) ? ~0ULL : mask_in; // mixed: only the first wave of each SIMD is sparse
    extern __shared__ unsigned char lds[];
    float a[16];
    double d[16];
    const float x = seed + 1.0f, y = seed * 0.5f + 0.25f;
    const double xd = (double)seed + 1.0, yd = (double)seed * 0.5 + 0.25;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + (float)(threadIdx.x + i);
        d[i] = (double)seed + (double)(threadIdx.x * 3 + i);
    }
    if (threadIdx.x == 1023) lds[0] = 1;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    if ((mask >> (threadIdx.x & 63)) & 1ULL) { // the loop runs with EXEC = mask
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
                    else if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(xd), "v"(yd));
                    else if (OP == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
                    else asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(a[i]) : "v"(x));
                }
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + (float)d[i];
    if (s == 1.2345f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[(size_t)blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*Kern)(unsigned long long*, float*, float, unsigned long long, int);

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int n_cu = prop.multiProcessorCount, k = 3;
    Kern kern[N_OPS] = {k_masked<FMA_F32>, k_masked<FMA_F64>, k_masked<CVT_F32_U32>, k_masked<ALIGNBIT>};
    unsigned long long* d_cyc = nullptr;
    float* d_sink = nullptr;
    if (hipMalloc(&d_cyc, sizeof(unsigned long long) * (size_t)n_cu * 16) != hipSuccess || hipMalloc(&d_sink, 16) != hipSuccess) return 1;
    struct M { char name[48]; unsigned long long mask; };
    std::vector<M> masks;
    auto add = [&](const char* fmt, int n, unsigned long long m) {
        M e;
        std::snprintf(e.name, sizeof(e.name), fmt, n);
        e.mask = m;
        masks.push_back(e);
    };
    for (int n : {64, 48, 32, 24, 20, 16, 12, 8, 4, 2, 1}) add("first %d lanes", n, n == 64 ? ~0ULL : ((1ULL << n) - 1ULL));
    // the same numbers of lanes spread evenly over the wave (lane 0 always active: it holds the clock)
    for (int n : {32, 16, 8, 4, 2}) {
        unsigned long long m = 0;
        for (int i = 0; i < n; ++i) m |= 1ULL << (i * (64 / n));
        add("%d lanes spread evenly", n, m);
    }
    add("lanes 0-15 and 32-47 (%d)", 32, 0x0000ffff0000ffffULL);
    add("one lane in each 16-lane row (%d)", 4, 0x0001000100010001ULL);
    add("lanes 0-7 of each 16-lane row (%d)", 32, 0x00ff00ff00ff00ffULL);
    add("lanes 0-3 of each 16-lane row (%d)", 16, 0x000f000f000f000fULL);
    std::printf("{\n \"device\": \"%s\", \"unit\": \"shader cycles per wave64 instruction per SIMD (lane 0 is active under every mask and holds the clock)\",\n", prop.gcnArchName);
    auto run = [&](int op, int kk, unsigned long long mask, int mixed, double* sparse_wave, double* dense_wave) -> double {
        const int lds = 160 * 1024 - 1024;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern[op]), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(kern[op], dim3(n_cu), dim3(256 * kk), lds, 0, d_cyc, d_sink, 0.0f, mask, mixed);
            if (hipDeviceSynchronize() != hipSuccess) return -1;
            std::vector<unsigned long long> all((size_t)n_cu * 16);
            if (hipMemcpy(all.data(), d_cyc, all.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
            std::vector<unsigned long long> worst, sp, de;
            for (int b = 0; b < n_cu; ++b)
                for (int sd = 0; sd < 4; ++sd) {
                    unsigned long long m = 0;
                    for (int w = sd; w < 4 * kk; w += 4) {
                        m = std::max(m, all[(size_t)b * 16 + w]);
                        (w < 4 ? sp : de).push_back(all[(size_t)b * 16 + w]);
                    }
                    worst.push_back(m);
                }
            std::sort(worst.begin(), worst.end());
            const double v = (double)worst[worst.size() / 2];
            if (v < best) {
                best = v;
                std::sort(sp.begin(), sp.end());
                std::sort(de.begin(), de.end());
                if (sparse_wave) *sparse_wave = (double)sp[sp.size() / 2] / (ITER * 128.0);
                if (dense_wave && !de.empty()) *dense_wave = (double)de[de.size() / 2] / (ITER * 128.0);
            }
        }
        return best / ((double)kk * ITER * 128);
    };
    std::printf(" \"three waves per SIMD, every wave under the mask\": {\n");
    for (int op = 0; op < N_OPS; ++op) {
        std::printf("  \"%s\": {", kNames[op]);
        for (size_t mi = 0; mi < masks.size(); ++mi) std::printf("%s\"%s\": %.3f", mi ? ", " : "", masks[mi].name, run(op, 3, masks[mi].mask, 0, nullptr, nullptr));
        std::printf("}%s\n", op + 1 < N_OPS ? "," : "");
    }
    std::printf(" },\n \"one wave per SIMD, under the mask\": {\n");
    for (int op = 0; op < N_OPS; ++op) {
        std::printf("  \"%s\": {", kNames[op]);
        int first = 1;
        for (int n : {64, 20, 16, 8, 1}) {
            std::printf("%s\"first %d lanes\": %.3f", first ? "" : ", ", n, run(op, 1, n == 64 ? ~0ULL : ((1ULL << n) - 1ULL), 0, nullptr, nullptr));
            first = 0;
        }
        std::printf("}%s\n", op + 1 < N_OPS ? "," : "");
    }
    std::printf(" },\n \"three waves per SIMD, ONE under the mask and two with all lanes: cycles per instruction as each kind of wave sees them (wave time / its instructions)\": {\n");
    for (int op = 0; op < N_OPS; ++op) {
        std::printf("  \"%s\": {", kNames[op]);
        int first = 1;
        for (int n : {64, 16, 8, 1}) {
            double sw = 0, dw = 0;
            run(op, 3, n == 64 ? ~0ULL : ((1ULL << n) - 1ULL), 1, &sw, &dw);
            std::printf("%s\"first %d lanes\": {\"masked wave\": %.3f, \"full waves\": %.3f}", first ? "" : ", ", n, sw, dw);
            first = 0;
        }
        std::printf("}%s\n", op + 1 < N_OPS ? "," : "");
    }
    std::printf(" }\n}\n");
    return 0;
}

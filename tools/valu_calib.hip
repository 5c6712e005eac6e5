// Developer tool: measures what one SIMD of MI355X issues per cycle for the vector instructions K1 / K3 are made of, at
// 1, 2, 3 and 4 resident waves per SIMD — the denominator of the `valu` roofline in bench.py (profiles/r03_valu_calibration.json).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_calib.hip -o tools/valu_calib && tools/valu_calib > profiles/r03_valu_calibration.json
// Every kernel runs ITER x 128 INDEPENDENT instructions of one kind (16 accumulators, so no instruction waits for the one
// before it) per wave, brackets them with s_memtime (shader cycles) and reports wave-instructions per cycle per SIMD =
// waves per SIMD x instructions / cycles.  k waves per SIMD = ONE workgroup of 256 k threads per CU (its waves are dealt
// round-robin over the CU's four SIMDs), pinned by giving the workgroup all of the CU's LDS and launching exactly one per CU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));      \
            return 1;                                                         \
        }                                                                     \
    } while (0)

constexpr int ITER = 2048;

enum Op : int {
    FMA_F32, PK_FMA_F32, FMA_F64, MUL_F64, ADD_F64, CVT_F32_U32, CVT_F32_U32_SDWA, ALIGNBIT, MAX_F32, MAX3_F32, CNDMASK, CMP_LE_F32,
    MUL_LO_U32, XOR_B32, MOV_B32, MOV_B64, ADD_U32, LSHL_ADD_U64, MAD_U64_U32, READLANE, WRITELANE, RCP_F32, RSQ_F64, RCP_F64,
    CVT_F64_F32, CVT_F32_F64, CMP_LT_F64, PERM_B32, AND_OR_B32, AND_B32, OR_B32, LSHRREV_B32, BFE_U32, MIN_F32, MUL_F32, ADD_F32, MIN3_F32, MED3_F32, CNDMASK_VCC, LSHL_OR_B32, CVT_UBYTE0, SUB_F32, FMAC_F32, CNDMASK_VCC_E64, CNDMASK_VCC_MIX, MAX_U32, N_OPS
};
static const char* kNames[N_OPS] = {
    "v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f32_u32", "v_cvt_f32_u32_sdwa", "v_alignbit_b32", "v_max_f32",
    "v_max3_f32", "v_cndmask_b32", "v_cmp_le_f32", "v_mul_lo_u32", "v_xor_b32", "v_mov_b32", "v_mov_b64", "v_add_u32", "v_lshl_add_u64",
    "v_mad_u64_u32", "v_readlane_b32", "v_writelane_b32", "v_rcp_f32", "v_rsq_f64", "v_rcp_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_cmp_lt_f64",
    "v_perm_b32", "v_and_or_b32", "v_and_b32", "v_or_b32", "v_lshrrev_b32", "v_bfe_u32", "v_min_f32", "v_mul_f32", "v_add_f32", "v_min3_f32", "v_med3_f32",
    "v_cndmask_b32_vcc", "v_lshl_or_b32", "v_cvt_f32_ubyte0", "v_sub_f32", "v_fmac_f32", "v_cndmask_b32_e64_vcc", "v_cndmask_b32_vcc+v_fma_f32 (pair)", "v_max_u32"};

// one instruction of kind OP on accumulator `a` (32-bit) / `d` (64-bit); x, y, xd, yd are loop-invariant operands
#define ONE(OP, a, d)                                                                                                  \
    do {                                                                                                               \
        if (OP == FMA_F32) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));                        \
        else if (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(xd), "v"(yd));           \
        else if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d) : "v"(xd), "v"(yd));                 \
        else if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d) : "v"(xd));                              \
        else if (OP == ADD_F64) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d) : "v"(xd));                              \
        else if (OP == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a));                                    \
        else if (OP == CVT_F32_U32_SDWA) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a)); \
        else if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(a) : "v"(x));                     \
        else if (OP == MAX_F32) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a) : "v"(x));                               \
        else if (OP == MAX3_F32) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));                 \
        else if (OP == CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(a) : "v"(x));            \
        else if (OP == CMP_LE_F32) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a), "v"(x) : "vcc");                \
        else if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(x));                         \
        else if (OP == XOR_B32) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a) : "v"(x));                               \
        else if (OP == MOV_B32) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(x));                                   \
        else if (OP == MOV_B64) asm volatile("v_mov_b64 %0, %1" : "=v"(d) : "v"(xd));                                  \
        else if (OP == ADD_U32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a) : "v"(x));                               \
        else if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(d) : "v"(xd));                 \
        else if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d) : "v"(x), "v"(y) : "vcc"); \
        else if (OP == READLANE) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a) : "s20");                         \
        else if (OP == WRITELANE) asm volatile("v_writelane_b32 %0, s21, 3" : "+v"(a) : : "s21");                      \
        else if (OP == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a));                                            \
        else if (OP == RSQ_F64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d));                                            \
        else if (OP == RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d));                                            \
        else if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a));                           \
        else if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a) : "v"(d));                           \
        else if (OP == CMP_LT_F64) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d), "v"(xd) : "vcc");               \
        else if (OP == PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));                 \
        else if (OP == AND_OR_B32) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));             \
        else if (OP == AND_B32) asm volatile("v_and_b32 %0, %1, %0" : "+v"(a) : "v"(x));                               \
        else if (OP == OR_B32) asm volatile("v_or_b32 %0, %1, %0" : "+v"(a) : "v"(x));                                 \
        else if (OP == LSHRREV_B32) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a));                                 \
        else if (OP == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 16" : "+v"(a));                                     \
        else if (OP == MIN_F32) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a) : "v"(x));                               \
        else if (OP == MUL_F32) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "v"(x));                               \
        else if (OP == ADD_F32) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "v"(x));                               \
        else if (OP == MIN3_F32) asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));                 \
        else if (OP == MED3_F32) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y));                 \
        else if (OP == CNDMASK_VCC) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(x) : );               \
        else if (OP == LSHL_OR_B32) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a) : "v"(x));                    \
        else if (OP == CVT_UBYTE0) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a));                                  \
        else if (OP == SUB_F32) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a) : "v"(x));                               \
        else if (OP == FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(x), "v"(y));                     \
        else if (OP == CNDMASK_VCC_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a) : "v"(x));          \
        else if (OP == CNDMASK_VCC_MIX) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_fma_f32 %2, %1, %1, %2" : "+v"(a), "+v"(b2) : "v"(x)); \
        else if (OP == MAX_U32) asm volatile("v_max_u32 %0, %1, %0" : "+v"(a) : "v"(x));                               \
    } while (0)

template <int OP>
__global__ __launch_bounds__(1024) void k_issue(unsigned long long* cycles, float* sink, float seed) {
    extern __shared__ unsigned char lds[]; // only to pin the number of workgroups a CU holds
    float a[16];
    double d[16];
    const float x = seed + 1.0f, y = seed * 0.5f + 0.25f;
    float b2 = seed;
    const double xd = (double)seed + 1.0, yd = (double)seed * 0.5 + 0.25;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        a[i] = seed + (float)(threadIdx.x + i);
        d[i] = (double)seed + (double)(threadIdx.x * 3 + i);
    }
    if (threadIdx.x == 1023) lds[0] = 1;
    asm volatile("s_mov_b64 s[22:23], 0x5555\n s_mov_b64 vcc, 0x3333" : : : "s22", "s23", "vcc"); // the lane masks v_cndmask selects by
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) ONE(OP, a[i], d[i]);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + (float)d[i];
    s += b2;
    if (s == 1.2345f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cycles[(size_t)blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*Kern)(unsigned long long*, float*, float);
template <int OP>
static void fill(Kern* k) {
    k[OP] = k_issue<OP>;
    if constexpr (OP + 1 < N_OPS) fill<OP + 1>(k);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    Kern kern[N_OPS];
    fill<0>(kern);
    unsigned long long* d_cyc = nullptr;
    float* d_sink = nullptr;
    CK(hipMalloc(&d_cyc, sizeof(unsigned long long) * (size_t)n_cu * 16));
    CK(hipMalloc(&d_sink, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::printf("{\n \"device\": \"%s\", \"compute_units\": %d, \"simds\": %d, \"instructions_per_wave\": %d,\n", prop.gcnArchName, n_cu, n_cu * 4, ITER * 128);
    std::printf(" \"unit\": \"wave64 instructions per shader cycle per SIMD (s_memtime ticks inside the kernel); clock_ghz = those ticks / HIP-event time\",\n \"ops\": {\n");
    for (int op = 0; op < N_OPS; ++op) {
        std::printf("  \"%s\": {", kNames[op]);
        for (int k = 1; k <= 4; ++k) {
            const int lds = 160 * 1024 - 1024; // the whole CU: one workgroup per CU
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern[op]), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
                std::fprintf(stderr, "cannot request %d bytes of LDS per workgroup\n", lds);
                return 1;
            }
            const int grid = n_cu, block = 256 * k;
            float best_ms = 1e30f;
            std::vector<unsigned long long> cyc((size_t)grid * 4 * k);
            double med = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern[op], dim3(grid), dim3(block), lds, 0, d_cyc, d_sink, 0.0f);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best_ms) {
                    best_ms = ms;
                    std::vector<unsigned long long> all((size_t)n_cu * 16);
                    CK(hipMemcpy(all.data(), d_cyc, all.size() * 8, hipMemcpyDeviceToHost));
                    cyc.clear();
                    for (int b = 0; b < grid; ++b)
                        for (int w = 0; w < 4 * k; ++w) cyc.push_back(all[(size_t)b * 16 + w]);
                    std::sort(cyc.begin(), cyc.end());
                    // per SIMD: the waves of one SIMD share it unevenly (the older wave issues first), so what the SIMD issued
                    // per cycle is all its waves' instructions over the LONGEST of their times; median over the SIMDs
                    std::vector<unsigned long long> worst;
                    for (int b = 0; b < grid; ++b)
                        for (int sd = 0; sd < 4; ++sd) {
                            unsigned long long m = 0;
                            for (int w = sd; w < 4 * k; w += 4) m = std::max(m, all[(size_t)b * 16 + w]);
                            worst.push_back(m);
                        }
                    std::sort(worst.begin(), worst.end());
                    med = (double)worst[worst.size() / 2];
                }
            }
            const double per_cycle = (double)k * ITER * 128 / med;
            std::printf("%s\"w%d\": {\"per_cycle_per_simd\": %.4f, \"cycles_per_instruction\": %.3f, \"clock_ghz\": %.3f, \"wave_time_max_over_min\": %.3f}", k == 1 ? "" : ", ", k,
                        per_cycle, 1.0 / per_cycle, med / (best_ms * 1e6), (double)cyc.back() / (double)cyc.front());
        }
        std::printf("}%s\n", op + 1 < N_OPS ? "," : "");
    }
    std::printf(" }\n}\n");
    return 0;
}

// MFMA || VALU co-execution microbenchmark for gfx950 (measurement tool, not part of libign_hip.so).
// Design question (DESIGN.md "two experts, two pipes"): the shapelet kernels are fp32-VALU bound, the FCN convolutions
// fp32-MFMA bound.  How much of one hides under the other when both run at once on two HIP streams, as a function of
// how many waves each kernel keeps resident per SIMD?
//   mfma kernel: 256-thread workgroups, each wave a chain of v_mfma_f32_32x32x2_f32 over 4 accumulators (no memory).
//   valu kernel: 64-thread workgroups, each wave 8 independent v_add_f32 chains (or the v_cmpx backward pattern).
// Occupancy of either kernel is capped with a dummy dynamic-LDS request.
// Build: hipcc -O3 --offload-arch=gfx950 coissue_microbench.hip -o coissue_microbench ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ int g_prio;
__global__ void __launch_bounds__(256) mfma_kernel(float* out, int iters, float seed) {
    extern __shared__ float dummy[];
    if (seed < 0.f) __builtin_amdgcn_s_setprio(3);
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = seed + r; c1[r] = seed - r; c2[r] = seed * r; c3[r] = 1.f + r; }
    const float a = threadIdx.x * 1e-6f + seed * 1e-3f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f) out[threadIdx.x] = s + dummy[0];
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(256) mfma_bf16_kernel(float* out, int iters, float seed) {
    extern __shared__ float dummy[];
    if (seed < 0.f) __builtin_amdgcn_s_setprio(3);
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = seed + r; c1[r] = seed - r; c2[r] = seed * r; c3[r] = 1.f + r; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + seed * 1e-2f * i); b[i] = (__bf16)(1.0f - threadIdx.x * 1e-3f * i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f) out[threadIdx.x] = s + dummy[0];
}

#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void __launch_bounds__(64) valu_kernel(float* out, int iters, float seed) {
    extern __shared__ float dummy[];
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float x0 = a0 * 0.5f, x1 = a1 * 0.5f, x2 = a2 * 0.25f, x3 = a3 * 0.125f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 64 x v_add_f32
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x0));)
        } else {                  // 16 x (v_cmpx ; v_add under exec ; s_mov exec,-1): the shapelet backward step (32 VALU)
            REP8(asm volatile("v_cmpx_gt_f32 %2, %3\n v_add_f32 %0, %0, %4\n s_mov_b64 exec, -1\n"
                              "v_cmpx_gt_f32 %5, %3\n v_add_f32 %1, %1, %4\n s_mov_b64 exec, -1\n"
                              : "+v"(a0), "+v"(a1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");)
        }
    }
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678f) out[threadIdx.x] = s + dummy[0];
}

static float g_seed = 1.5f;      // negative: the MFMA kernel raises its wave priority (s_setprio 3)
static float run(hipStream_t sm, hipStream_t sv, float* out, int mfma_wg, int mfma_iters, size_t mfma_lds, int valu_wg,
                 int valu_iters, size_t valu_lds, int mode, int order, float* t_m, float* t_v, int bf16 = 0) {
    hipEvent_t m0, m1, v0, v1, w0, w1;
    CHECK(hipEventCreate(&m0)); CHECK(hipEventCreate(&m1)); CHECK(hipEventCreate(&v0)); CHECK(hipEventCreate(&v1));
    CHECK(hipEventCreate(&w0)); CHECK(hipEventCreate(&w1));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(w0, sm));
    CHECK(hipStreamWaitEvent(sv, w0, 0));
    auto lm = [&]() {
        if (mfma_wg <= 0) return;
        CHECK(hipEventRecord(m0, sm));
        if (bf16) hipLaunchKernelGGL(mfma_bf16_kernel, dim3(mfma_wg), dim3(256), mfma_lds, sm, out, mfma_iters, g_seed);
        else hipLaunchKernelGGL(mfma_kernel, dim3(mfma_wg), dim3(256), mfma_lds, sm, out, mfma_iters, g_seed);
        CHECK(hipEventRecord(m1, sm));
    };
    auto lv = [&]() {
        if (valu_wg <= 0) return;
        CHECK(hipEventRecord(v0, sv));
        if (mode == 0) hipLaunchKernelGGL(valu_kernel<0>, dim3(valu_wg), dim3(64), valu_lds, sv, out, valu_iters, 1.5f);
        else hipLaunchKernelGGL(valu_kernel<1>, dim3(valu_wg), dim3(64), valu_lds, sv, out, valu_iters, 1.5f);
        CHECK(hipEventRecord(v1, sv));
    };
    if (order == 0) { lm(); lv(); } else { lv(); lm(); }
    CHECK(hipEventRecord(w1, sv));
    CHECK(hipStreamWaitEvent(sm, w1, 0));
    hipEvent_t end; CHECK(hipEventCreate(&end));
    CHECK(hipEventRecord(end, sm));
    CHECK(hipEventSynchronize(end));
    CHECK(hipDeviceSynchronize());
    float wall = 0.f;
    *t_m = *t_v = 0.f;
    CHECK(hipEventElapsedTime(&wall, w0, end));
    if (mfma_wg > 0) CHECK(hipEventElapsedTime(t_m, m0, m1));
    if (valu_wg > 0) CHECK(hipEventElapsedTime(t_v, v0, v1));
    return wall;
}

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == 1) { g_seed = -1.5f; printf("# MFMA kernel at s_setprio 3\n"); }
    float* out;
    CHECK(hipMalloc(&out, 1 << 20));
    hipStream_t sm, sv;
    CHECK(hipStreamCreateWithFlags(&sm, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    const int CUS = 256;
    float tm, tv;
    // warm-up
    run(sm, sv, out, CUS, 100, 0, CUS * 16, 100, 0, 0, 0, &tm, &tv);
    printf("# mfma: wg/CU x iters ; valu: waves/CU (LDS-capped) ; order 0 = mfma first\n");
    const int mi = 2000;                         // 2000 iters x 16 MFMA x 64 cyc = 2.05 M cycles ~ 0.95 ms at one wave/SIMD
    const int vi = 4000;                         // per wave 4000 x 64 v_add
    for (int bf16 = 0; bf16 < 2; ++bf16)
    for (int mode = 0; mode < 2; ++mode) {
        for (int mwg = 1; mwg <= 2; ++mwg) {
            const int caps[3] = {8, 16, 28};
            for (int ci = 0; ci < 3; ++ci) {
                const int cap = caps[ci];
                const size_t vlds = (size_t)(160 * 1024 / cap) & ~(size_t)255;
                const int vwg = CUS * cap * 2;   // two rounds of resident waves
                const int viters = mode == 0 ? vi : vi * 2;
                const int miters = mi * (mwg == 1 ? 2 : 1);
                float a_m, a_v, dummy;
                const float wm = run(sm, sv, out, CUS * mwg, miters, 0, 0, 0, 0, mode, 0, &a_m, &dummy, bf16);
                const float wv = run(sm, sv, out, 0, 0, 0, vwg, viters, vlds, mode, 0, &dummy, &a_v, bf16);
                for (int order = 0; order < 2; ++order) {
                    const float wb = run(sm, sv, out, CUS * mwg, miters, 0, vwg, viters, vlds, mode, order, &tm, &tv, bf16);
                    printf("%s valu-mode %d | mfma %d wg/CU alone %.3f ms | valu cap %2d waves/CU alone %.3f ms | both(order %d) wall %.3f ms "
                           "(mfma %.3f valu %.3f)  sum-alone %.3f  hidden %.0f%% of the shorter\n", bf16 ? "bf16-mfma" : "f32-mfma ",
                           mode, mwg, wm, cap, wv, order, wb, tm, tv, wm + wv, 100.f * (wm + wv - wb) / fminf(wm, wv));
                }
            }
        }
    }
    return 0;
}

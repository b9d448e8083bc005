// Same-wave MFMA + VALU interleave microbenchmark for gfx950 (measurement tool, not part of libign_hip.so).
// coissue_microbench.hip showed that a VALU-bound KERNEL starves next to an MFMA-bound kernel (different waves).  Question
// here: inside ONE wave, do independent VALU instructions issue in the shadow of that wave's own in-flight MFMAs?  Each wave
// runs, per iteration, NM independent v_mfma_f32_32x32x16_bf16 (4 accumulators) and NV independent v_add_f32 (8 chains),
// interleaved in program order; compared with the MFMA-only and VALU-only forms of the same loop at 1 / 2 / 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 samewave_microbench.hip -o samewave_microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define VADD8 asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n" \
                           "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n" \
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x0));

// MODE 0: MFMA only; 1: VALU only; 2: interleaved (per MFMA: VPM groups of 8 v_add)
template <int MODE, int VPM>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    extern __shared__ float dummy[];
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = seed + r; c1[r] = seed - r; c2[r] = seed * r; c3[r] = 1.f + r; }
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 1e-3f + seed * 1e-2f * i); b[i] = (__bf16)(1.0f - threadIdx.x * 1e-3f * i); }
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
          a7 = a0 + 7.f, x0 = a0 * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (MODE != 1) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            if (MODE != 0) { for (int q = 0; q < VPM; ++q) { VADD8 } }
            if (MODE != 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
            if (MODE != 0) { for (int q = 0; q < VPM; ++q) { VADD8 } }
            if (MODE != 1) c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
            if (MODE != 0) { for (int q = 0; q < VPM; ++q) { VADD8 } }
            if (MODE != 1) c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
            if (MODE != 0) { for (int q = 0; q < VPM; ++q) { VADD8 } }
        }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f) out[threadIdx.x] = s + dummy[0];
}

template <int MODE, int VPM>
static float run(float* out, int wgs, int iters, size_t lds) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, VPM>), dim3(wgs), dim3(256), lds, 0, out, 10, 1.5f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE, VPM>), dim3(wgs), dim3(256), lds, 0, out, iters, 1.5f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template <int VPM>
static void sweep(float* out) {
    const int CUS = 256, iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {                      // waves per SIMD = workgroups per CU (256 threads = 1 wave / SIMD)
        const size_t lds = (size_t)(160 * 1024 / wps) & ~(size_t)255;
        const float tm = run<0, VPM>(out, CUS * wps, iters, lds);
        const float tv = run<1, VPM>(out, CUS * wps, iters, lds);
        const float tb = run<2, VPM>(out, CUS * wps, iters, lds);
        printf("v_add per MFMA %3d | %d wave(s)/SIMD | mfma only %.3f ms | valu only %.3f ms | interleaved %.3f ms | sum %.3f max %.3f | "
               "hidden %.0f%% of the shorter\n", 8 * VPM, wps, tm, tv, tb, tm + tv, fmaxf(tm, tv), 100.f * (tm + tv - tb) / fminf(tm, tv));
    }
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 1 << 20));
    sweep<1>(out);      //  8 v_add (32 issue cycles) per 32-cycle MFMA
    sweep<2>(out);      // 16
    sweep<4>(out);      // 32
    return 0;
}

// VALU issue-rate microbenchmark for gfx950 (measurement tool, not part of libign_hip.so).
// Answers the design questions of the shapelet kernels (DESIGN.md "VALU ceiling"):
//   * wave-instructions per cycle per SIMD for plain v_add_f32 / v_sub+v_add|abs| / v_pk_add_f32,
//   * the cost of the cmp -> cndmask -> add select-accumulate (with the compiler's hazard s_nop),
//   * the v_cmpx + masked v_add + s_mov exec alternative (2 VALU + 1 SALU per element).
// Build: hipcc -O3 --offload-arch=gfx950 valu_microbench.hip -o valu_microbench ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2000;

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int MODE>
__global__ void __launch_bounds__(256) bench_kernel(float* out, float seed) {
    float a0 = threadIdx.x * 1e-3f + seed, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    float a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float x0 = a0 * 0.5f, x1 = a1 * 0.5f, x2 = a2 * 0.25f, x3 = a3 * 0.125f;
    float t0, t1, t2, t3;
    const float w = seed * 0.75f;
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {          // 64 x v_add_f32 (8 independent chains)
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x0));)
        } else if (MODE == 1) {   // 32 x (v_sub_f32 ; v_add_f32 |.|) = 64 VALU: the forward inner loop
            REP8(asm volatile("v_sub_f32 %4, %8, %9\n v_add_f32 %0, %0, |%4|\n v_sub_f32 %5, %10, %9\n v_add_f32 %1, %1, |%5|\n"
                              "v_sub_f32 %6, %11, %9\n v_add_f32 %2, %2, |%6|\n v_sub_f32 %7, %12, %9\n v_add_f32 %3, %3, |%7|\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                              : "v"(x0), "s"(w), "v"(x1), "v"(x2), "v"(x3));)
        } else if (MODE == 2) {   // 64 x v_pk_add_f32 (4 independent 2-wide chains)
            REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
                              : "v"(*(double*)&x0));)
        } else if (MODE == 3) {   // 16 x (v_cmp ; s_nop 0 ; v_cndmask ; v_add): the compiler's backward inner loop
            REP8(asm volatile("v_cmp_gt_f32 vcc, %4, %5\n s_nop 0\n v_cndmask_b32 %2, -%6, %6, vcc\n v_add_f32 %0, %0, %2\n"
                              "v_cmp_gt_f32 vcc, %7, %5\n s_nop 0\n v_cndmask_b32 %3, -%6, %6, vcc\n v_add_f32 %1, %1, %3\n"
                              : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");)
        } else if (MODE == 4) {   // 16 x (v_cmpx ; v_add under exec ; s_mov exec,-1): 2 VALU + 1 SALU per element
            REP8(asm volatile("v_cmpx_gt_f32 %2, %3\n v_add_f32 %0, %0, %4\n s_mov_b64 exec, -1\n"
                              "v_cmpx_gt_f32 %5, %3\n v_add_f32 %1, %1, %4\n s_mov_b64 exec, -1\n"
                              : "+v"(a0), "+v"(a1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");)
        } else if (MODE == 5) {   // software-pipelined select-accumulate: 2 compares into different SGPR pairs first
            REP8(asm volatile("v_cmp_gt_f32 s[20:21], %4, %5\n v_cmp_gt_f32 s[22:23], %7, %5\n"
                              "v_cndmask_b32 %2, -%6, %6, s[20:21]\n v_cndmask_b32 %3, -%6, %6, s[22:23]\n"
                              "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %3\n"
                              : "+v"(a0), "+v"(a1), "=&v"(t0), "=&v"(t1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3)
                              : "s20", "s21", "s22", "s23");)
        } else if (MODE == 6) {   // 64 x v_fma_f32
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                              "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x0));)
        } else if (MODE == 7) {   // 64 x v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                              "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                              : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6)
                              : "v"(*(double*)&x0));)
        } else if (MODE == 8) {   // 64 x v_sad_u32 acc, x, s_w, acc   (|x - w| + acc on 32-bit unsigned: ONE op per element)
            REP8(asm volatile("v_sad_u32 %0, %8, %9, %0\n v_sad_u32 %1, %10, %9, %1\n v_sad_u32 %2, %11, %9, %2\n v_sad_u32 %3, %12, %9, %3\n"
                              "v_sad_u32 %4, %8, %9, %4\n v_sad_u32 %5, %10, %9, %5\n v_sad_u32 %6, %11, %9, %6\n v_sad_u32 %7, %12, %9, %7\n"
                              : "+v"(*(unsigned*)&a0), "+v"(*(unsigned*)&a1), "+v"(*(unsigned*)&a2), "+v"(*(unsigned*)&a3),
                                "+v"(*(unsigned*)&a4), "+v"(*(unsigned*)&a5), "+v"(*(unsigned*)&a6), "+v"(*(unsigned*)&a7)
                              : "v"(*(unsigned*)&x0), "s"(w), "v"(*(unsigned*)&x1), "v"(*(unsigned*)&x2), "v"(*(unsigned*)&x3));)
        } else if (MODE == 9) {   // the same with a float flush every 16 elements: 64 v_sad + 4 x (v_cvt_f32_u32 ; v_add_f32)
            REP8(asm volatile("v_sad_u32 %0, %8, %9, 0\n v_sad_u32 %1, %10, %9, 0\n v_sad_u32 %2, %11, %9, 0\n v_sad_u32 %3, %12, %9, 0\n"
                              "v_sad_u32 %0, %10, %9, %0\n v_sad_u32 %1, %11, %9, %1\n v_sad_u32 %2, %12, %9, %2\n v_sad_u32 %3, %8, %9, %3\n"
                              : "+v"(*(unsigned*)&a0), "+v"(*(unsigned*)&a1), "+v"(*(unsigned*)&a2), "+v"(*(unsigned*)&a3),
                                "+v"(*(unsigned*)&a4), "+v"(*(unsigned*)&a5), "+v"(*(unsigned*)&a6), "+v"(*(unsigned*)&a7)
                              : "v"(*(unsigned*)&x0), "s"(w), "v"(*(unsigned*)&x1), "v"(*(unsigned*)&x2), "v"(*(unsigned*)&x3));)
            asm volatile("v_cvt_f32_u32 %4, %0\n v_add_f32 %8, %8, %4\n v_cvt_f32_u32 %5, %1\n v_add_f32 %9, %9, %5\n"
                         "v_cvt_f32_u32 %6, %2\n v_add_f32 %10, %10, %6\n v_cvt_f32_u32 %7, %3\n v_add_f32 %11, %11, %7\n"
                         : "+v"(*(unsigned*)&a0), "+v"(*(unsigned*)&a1), "+v"(*(unsigned*)&a2), "+v"(*(unsigned*)&a3),
                           "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

struct Mode { const char* name; int valu_per_rep8; };

template <int MODE>
double run(float* out, int blocks, int threads) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(bench_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f + r);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 5.0;
}

int main() {
    float* out;
    const int threads = 256;
    CHECK(hipMalloc(&out, (size_t)256 * 16 * threads * sizeof(float)));
    // VALU instructions per iteration of the REP8 body
    const int valu[10] = {64, 64, 64, 48, 32, 48, 64, 64, 64, 72};
    const char* names[10] = {"v_add_f32", "v_sub+v_add|abs| (fwd loop)", "v_pk_add_f32", "cmp/nop/cndmask/add (bwd loop)",
                            "cmpx/add/s_mov exec", "2x cmp(sgpr) / 2x cndmask / 2x add", "v_fma_f32", "v_pk_fma_f32",
                            "v_sad_u32 (|x-w|+acc, 1 op/elem)", "v_sad_u32 x16 + cvt/add flush"};
    const int elems[10] = {64, 32, 128, 16, 16, 16, 64, 128, 64, 64};   // useful per-lane results per iteration
    for (int wps = 1; wps <= 8; wps *= 2) {                    // waves per SIMD
        const int blocks = 256 * wps;                          // 256 CUs x wps blocks of 4 waves
        printf("--- %d wave(s) per SIMD (%d blocks x %d threads)\n", wps, blocks, threads);
        double ms[10];
        ms[0] = run<0>(out, blocks, threads); ms[1] = run<1>(out, blocks, threads); ms[2] = run<2>(out, blocks, threads);
        ms[3] = run<3>(out, blocks, threads); ms[4] = run<4>(out, blocks, threads); ms[5] = run<5>(out, blocks, threads);
        ms[6] = run<6>(out, blocks, threads); ms[7] = run<7>(out, blocks, threads);
        ms[8] = run<8>(out, blocks, threads); ms[9] = run<9>(out, blocks, threads);
        for (int m = 0; m < 10; ++m) {
            const double waves = (double)blocks * threads / 64.0;
            const double winst = waves * ITERS * valu[m];
            const double per_simd_per_us = winst / 1024.0 / (ms[m] * 1e3);     // wave-instr / us / SIMD
            const double gelem = waves * 64.0 * ITERS * elems[m] / (ms[m] * 1e-3) / 1e12;
            printf("  %-40s %8.3f ms  %7.1f wave-VALU/us/SIMD (=%5.2f per cycle @2.4GHz)  %7.2f T useful lane-results/s\n",
                   names[m], ms[m], per_simd_per_us, per_simd_per_us / 2400.0, gelem);
        }
    }
    return 0;
}

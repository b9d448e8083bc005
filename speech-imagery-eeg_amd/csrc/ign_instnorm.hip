// Instance normalisation fused with the (B,T,C) -> (B,C,T) transpose.
// Replaces IGN/model/Shapelet.py:186-187.  HBM-bound: 4*B*T*C bytes in, 4*B*C*T out (x2 with xt).
// A block owns (b, CT channels): the T x CT tile is staged once in LDS (row pitch CT+1: column reads are
// conflict-free), mean and unbiased variance are two passes over LDS (no E[x^2]-E[x]^2 cancellation: EEG
// arrives in microvolts with large offsets), and the normalised rows leave as coalesced (c, t) lines.
#include "ign_common.h"

__global__ void __launch_bounds__(256) instnorm_kernel(const float* __restrict__ x, float* __restrict__ xn,
                                                       float* __restrict__ xt, int B, int T, int C, int CT,
                                                       float eps) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int pitch = CT + 1;
    const int nct = (C + CT - 1) / CT;
    const int b = blockIdx.x / nct;
    const int c0 = (blockIdx.x - b * nct) * CT;
    const int tid = threadIdx.x;
    const float* xb = x + (size_t)b * T * C;
    for (int idx = tid; idx < T * CT; idx += 256) {
        const int t = idx / CT, cc = idx - t * CT;
        const int c = c0 + cc;
        tile[t * pitch + cc] = (c < C) ? xb[(size_t)t * C + c] : 0.f;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    for (int cc = wave; cc < CT; cc += 4) {
        const int c = c0 + cc;
        if (c >= C) break;
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s += tile[t * pitch + cc];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = s / (float)T;
        float v = 0.f;
        for (int t = lane; t < T; t += 64) {
            const float dv = tile[t * pitch + cc] - mean;
            v = fmaf(dv, dv, v);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        const float denom = sqrtf(v / (float)(T - 1)) + eps;      // torch.std: unbiased; eps outside the sqrt
        float* on = xn + ((size_t)b * C + c) * T;
        float* ot = xt ? xt + ((size_t)b * C + c) * T : nullptr;
        for (int t = lane; t < T; t += 64) {
            const float xv = tile[t * pitch + cc];
            on[t] = (xv - mean) / denom;
            if (ot) ot[t] = xv;
        }
    }
}

extern "C" int ign_instnorm_fwd(const float* x_btc, float* xn_bct, float* xt_bct, int B, int T, int C, float eps,
                                void* stream) {
    if (!x_btc || !xn_bct || B <= 0 || T <= 0 || C <= 0) {
        ign_set_error("ign_instnorm_fwd: null pointer or non-positive dimension (B=%d T=%d C=%d)", B, T, C);
        return IGN_E_ARG;
    }
    int CT = 16;
    while (CT > 1 && (size_t)T * (CT + 1) * 4 > 64 * 1024) CT >>= 1;
    const size_t lds = (size_t)T * (CT + 1) * 4;
    if (lds > 160 * 1024) {
        ign_set_error("ign_instnorm_fwd: T=%d does not fit the LDS tile", T);
        return IGN_E_TOOBIG;
    }
    const int nct = (C + CT - 1) / CT;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)instnorm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    IgnScopedTimer tm("instnorm", (hipStream_t)stream);
    hipLaunchKernelGGL(instnorm_kernel, dim3((unsigned)B * nct), dim3(256), lds, (hipStream_t)stream, x_btc, xn_bct,
                       xt_bct, B, T, C, CT, eps);
    return ign_check_launch("instnorm_kernel");
}

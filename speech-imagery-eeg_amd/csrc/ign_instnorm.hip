// Instance normalisation fused with the (B,T,C) -> (B,C,T) transpose.
// Replaces IGN/model/Shapelet.py:186-187.  HBM-bound: 4*B*T*C bytes in, 4*B*C*T out (x2 with xt).
// A block owns (b, CT channels): the T x CT tile is staged once in LDS (row pitch CT+1: column reads are
// conflict-free), mean and unbiased variance are two passes over LDS (no E[x^2]-E[x]^2 cancellation: EEG
// arrives in microvolts with large offsets), and the normalised rows leave as coalesced (c, t) lines.
// CT = 32 channels wherever the tile fits 150 KB of LDS (T <= 1160): the input rows are then read in 128-byte segments.  With
// CT = 8 (36 KB tiles) PMC showed 2.4x the algorithmic fetch traffic -- 32-byte pieces of 128-byte lines -- and 115 us for the
// 250 MB; the large tile uses 1024-thread blocks so that one block per CU still keeps 32 KB of loads in flight.
#include "ign_common.h"

__global__ void __launch_bounds__(1024) instnorm_kernel(const float* __restrict__ x, float* __restrict__ xn,
                                                       float* __restrict__ xt, int B, int T, int C, int CT,
                                                       float eps, float* __restrict__ amax_slot) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int pitch = CT + 1;
    const int nct = (C + CT - 1) / CT;
    const int b = blockIdx.x / nct;
    const int c0 = (blockIdx.x - b * nct) * CT;
    const int tid = threadIdx.x;
    const float* xb = x + (size_t)b * T * C;
    // eight loads in flight per thread before the first LDS store (a load -> store loop pays one memory round trip per
    // iteration: 62 of them for a 1000 x 16 tile)
    const int nthr = blockDim.x;
    for (int i0 = tid; i0 < T * CT; i0 += 8 * nthr) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = i0 + u * nthr;
            const int t = idx / CT, cc = idx - t * CT;
            v[u] = (idx < T * CT && c0 + cc < C) ? xb[(size_t)t * C + c0 + cc] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = i0 + u * nthr;
            const int t = idx / CT, cc = idx - t * CT;
            if (idx < T * CT) tile[t * pitch + cc] = v[u];
        }
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    float am = 0.f;                                               // max |x| of this wave's share of the tile (amax_slot only)
    for (int cc = wave; cc < CT; cc += (nthr >> 6)) {
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
            am = fmaxf(am, fabsf(xv));
        }
    }
    if (amax_slot) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
        if (lane == 0) ign_atomic_absmax(amax_slot, am);
    }
}

static int instnorm_impl(const char* who, const float* x_btc, float* xn_bct, float* xt_bct, int B, int T, int C, float eps,
                         float* amax_slot, void* stream) {
    if (!x_btc || !xn_bct || B <= 0 || T <= 0 || C <= 0) {
        ign_set_error("%s: null pointer or non-positive dimension (B=%d T=%d C=%d)", who, B, T, C);
        return IGN_E_ARG;
    }
    int CT = 32;
    while (CT > 1 && (size_t)T * (CT + 1) * 4 > 150 * 1024) CT >>= 1;
    const size_t lds = (size_t)T * (CT + 1) * 4;
    const int threads = lds > 64 * 1024 ? 1024 : 256;
    if (lds > 160 * 1024) {
        ign_set_error("%s: T=%d does not fit the LDS tile", who, T);
        return IGN_E_TOOBIG;
    }
    const int nct = (C + CT - 1) / CT;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)instnorm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    IgnScopedTimer tm("instnorm", (hipStream_t)stream);
    hipLaunchKernelGGL(instnorm_kernel, dim3((unsigned)B * nct), dim3(threads), lds, (hipStream_t)stream, x_btc, xn_bct,
                       xt_bct, B, T, C, CT, eps, amax_slot);
    return ign_check_launch("instnorm_kernel");
}

extern "C" int ign_instnorm_fwd(const float* x_btc, float* xn_bct, float* xt_bct, int B, int T, int C, float eps,
                                void* stream) {
    return instnorm_impl("ign_instnorm_fwd", x_btc, xn_bct, xt_bct, B, T, C, eps, nullptr, stream);
}

// The same pass also takes max |x| of the raw input (atomic maximum into *amax_slot, which the caller zeroed): the FCN expert's
// fp16 GEMMs scale their first operand by it (ign_clconv_fwd_h3), and the batch is in LDS here anyway.
extern "C" int ign_instnorm_fwd_amax(const float* x_btc, float* xn_bct, float* xt_bct, int B, int T, int C, float eps,
                                     float* amax_slot, void* stream) {
    if (!amax_slot) { ign_set_error("ign_instnorm_fwd_amax: null amax_slot"); return IGN_E_ARG; }
    return instnorm_impl("ign_instnorm_fwd_amax", x_btc, xn_bct, xt_bct, B, T, C, eps, amax_slot, stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// On-GPU input pipeline of the CHISCO loader: raw (B, C, T) microvolt recordings -> per-sample, per-channel standardised
// (B, T, C) batches, i.e. Normalizer('per_sample_std') of IGN/data_factory/eeg.py:332-367 followed by the (C,T) -> (T,C)
// item transpose of the UEA item contract, for a whole batch in two HBM passes instead of B CPU passes in the DataLoader.
//   pass 1: one wave per (b, c) row: mean and unbiased variance in two sweeps over the row held in registers / L1
//           (no E[x^2] - E[x]^2 cancellation: the recordings carry offsets of 1e4 uV);
//   pass 2: 64 (t) x 32 (c) tiles transposed through LDS: coalesced 256-byte reads along t, 128-byte writes along c.
__global__ void __launch_bounds__(256) std_rowstats_kernel(const float* __restrict__ x, float* __restrict__ stats, int rows,
                                                           int T, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (size_t)row * T;
    float s = 0.f;
    for (int t = lane; t < T; t += 64) s += xr[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)T;
    float v = 0.f;
    for (int t = lane; t < T; t += 64) {
        const float dv = xr[t] - mean;
        v = fmaf(dv, dv, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) {
        stats[2 * (size_t)row] = mean;
        stats[2 * (size_t)row + 1] = 1.f / (sqrtf(v / (float)(T - 1)) + eps);      // ddof = 1, eps outside the sqrt
    }
}

__global__ void __launch_bounds__(256) std_apply_transpose_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                  float* __restrict__ out, int C, int T) {
    __shared__ float tile[32][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int cc = w * 8 + i, c = c0 + cc, t = t0 + lane;
        float v = 0.f;
        if (c < C && t < T) {
            const size_t row = (size_t)b * C + c;
            v = (x[row * T + t] - stats[2 * row]) * stats[2 * row + 1];
        }
        tile[cc][lane] = v;
    }
    __syncthreads();
    const int cc = threadIdx.x & 31, tr = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int tl = tr * 8 + i, t = t0 + tl, c = c0 + cc;
        if (t < T && c < C) out[((size_t)b * T + t) * C + c] = tile[cc][tl];
    }
}

extern "C" int ign_standardise_nct_to_btc(const float* x_nct, float* out_btc, float* stats_ws, int B, int C, int T, float eps,
                                          void* stream) {
    if (!x_nct || !out_btc || !stats_ws || B <= 0 || C <= 0 || T <= 1) {
        ign_set_error("ign_standardise_nct_to_btc: null pointer or bad dimension (B=%d C=%d T=%d)", B, C, T);
        return IGN_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * C;
    IgnScopedTimer tm("standardise", s);
    hipLaunchKernelGGL(std_rowstats_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x_nct, stats_ws, rows, T, eps);
    int rc;
    if ((rc = ign_check_launch("std_rowstats_kernel"))) return rc;
    hipLaunchKernelGGL(std_apply_transpose_kernel, dim3((T + 63) / 64, (C + 31) / 32, B), dim3(256), 0, s, x_nct, stats_ws, out_btc,
                       C, T);
    return ign_check_launch("std_apply_transpose_kernel");
}

// ---------------------------------------------------------------------------------------------------------------------
// Plain (B, T, C) -> (B, C, T) transpose of a batch: the EEG-CNN baseline takes electrodes-first input (IGN/model/eegcnn.py:134)
// while the loader's item contract is time-first, so the reference's harness hands it `x.permute(0, 2, 1)`.  64 (t) x 32 (c) tiles
// through LDS: 128-byte reads along c, 256-byte writes along t (torch's strided copy of the 125 MB batch took 0.16 ms per step).
__global__ void __launch_bounds__(256) transpose_btc_bct_kernel(const float* __restrict__ x, float* __restrict__ out, int T, int C) {
    __shared__ float tile[64][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 64;
    const int cc = threadIdx.x & 31, tr = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int tl = tr * 8 + i, t = t0 + tl, c = c0 + cc;
        tile[tl][cc] = (t < T && c < C) ? x[((size_t)b * T + t) * C + c] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int cl = w * 8 + i, c = c0 + cl, t = t0 + lane;
        if (c < C && t < T) out[((size_t)b * C + c) * T + t] = tile[lane][cl];
    }
}

extern "C" int ign_transpose_btc_to_bct(const float* x_btc, float* out_bct, int B, int T, int C, void* stream) {
    if (!x_btc || !out_bct || B <= 0 || T <= 0 || C <= 0 || B > 65535) {
        ign_set_error("ign_transpose_btc_to_bct: null pointer or bad dimension (B=%d T=%d C=%d)", B, T, C);
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(transpose_btc_bct_kernel, dim3((T + 63) / 64, (C + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, x_btc, out_bct,
                       T, C);
    return ign_check_launch("transpose_btc_bct_kernel");
}

// EEG-CNN convolution block, the pieces around the depthwise temporal convolutions (IGN/model/eegcnn.py:67-108):
//
//   * channel contraction   u[b,o,t] = sum_c W[o,c] x[b,c,t]            (the depthwise SPATIAL conv :71,92 -- electrodes -> 64
//                                                                          maps -- and the POINTWISE conv :79,100)
//     and its weight gradient dW[o,c] = sum_{b,t} du[b,o,t] x[b,c,t];    the input gradient is the same contraction with W^T
//   * BatchNorm2d (batch statistics) + ELU + AvgPool(1,P) as ONE op:     statistics pass + apply pass forward,
//     sums pass + apply pass backward  (:72-74,93-95 and :80-82,101-103) -- the (B,64,T) activation is read twice per direction
//     and nothing but the pooled output is written.
//
// Layout: (B, channels, T) rows of time, the layout the reference's Conv2d sees and ign_dwconv1d_* work on.  All of this is
// HBM- / issue-light fp32 VALU work on 65 MB tensors (2e9 FMA per contraction): no MFMA reshaping -- the contraction has K = 122
// resp. 64 and keeps its weights in SGPRs (wave-uniform), the gradient keeps a 8 x 4 register tile per lane over LDS slices.
// Reductions are two-stage in a fixed order (double for the BatchNorm moments): bitwise reproducible, no float atomics.
#include <algorithm>
#include "ign_common.h"

typedef const __attribute__((address_space(4))) float* cfloat_p;

// ------------------------------------------------------------------------------------------------ channel contraction
// thread <-> one time step of one sample, 64 output accumulators in registers; W^T (Ci, 64) is read through the constant
// address space: 64 s_load'ed scalars per input channel are the SGPR operands of the 64 FMAs that x[b,c,t] feeds.
__global__ void __launch_bounds__(256) chan_contract_kernel(const float* __restrict__ x, const float* __restrict__ wt64,
                                                            float* __restrict__ u, int Ci, int Co, int T) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const bool ok = t < T;
    const float* xb = x + (size_t)b * Ci * T + (ok ? t : 0);
    const cfloat_p w = (cfloat_p)(uintptr_t)wt64;
    float acc[64];
#pragma unroll
    for (int o = 0; o < 64; ++o) acc[o] = 0.f;
    int c = 0;
    for (; c + 4 <= Ci; c += 4) {
        float xv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[q] = xb[(size_t)(c + q) * T];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o = 0; o < 64; ++o) acc[o] = fmaf(w[(c + q) * 64 + o], xv[q], acc[o]);
    }
    for (; c < Ci; ++c) {
        const float xv = xb[(size_t)c * T];
#pragma unroll
        for (int o = 0; o < 64; ++o) acc[o] = fmaf(w[c * 64 + o], xv, acc[o]);
    }
    if (ok) {
        float* ub = u + (size_t)b * Co * T + t;
#pragma unroll
        for (int o = 0; o < 64; ++o)
            if (o < Co) ub[(size_t)o * T] = acc[o];
    }
}

// dW[o,c] = sum_{b,t} du[b,o,t] x[b,c,t].  A block walks slices (b, 64 time steps): du (<=64 rows) and x (<=128 rows) of the
// slice go to LDS (row pitch 65: the 4 x-rows of a lane are c, c+32, c+64, c+96 -> conflict-free, the 8 du-rows are a
// half-wave broadcast), lane <-> (8 outputs og*8.., 4 inputs cg + 32 i) keeps 32 accumulators.  Partials per block, reduced in
// ascending block order afterwards.
constexpr int CW_TC = 64, CW_PITCH = 65;
__global__ void __launch_bounds__(256) chan_contract_bwd_w_kernel(const float* __restrict__ du, const float* __restrict__ x,
                                                                  float* __restrict__ part, int B, int Ci, int Co, int T) {
    __shared__ float dus[64 * CW_PITCH];
    __shared__ float xs[128 * CW_PITCH];
    const int tid = threadIdx.x;
    const int og = tid >> 5, cg = tid & 31;
    const int nchunk = (T + CW_TC - 1) / CW_TC;
    const int nslice = B * nchunk;
    float acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;
    for (int sl = blockIdx.x; sl < nslice; sl += gridDim.x) {
        const int b = sl / nchunk, t0 = (sl - b * nchunk) * CW_TC;
        __syncthreads();
        for (int i = tid; i < 64 * CW_TC; i += 256) {
            const int r = i >> 6, tt = i & 63;
            dus[r * CW_PITCH + tt] = (r < Co && t0 + tt < T) ? du[((size_t)b * Co + r) * T + t0 + tt] : 0.f;
        }
        for (int i = tid; i < 128 * CW_TC; i += 256) {
            const int r = i >> 6, tt = i & 63;
            xs[r * CW_PITCH + tt] = (r < Ci && t0 + tt < T) ? x[((size_t)b * Ci + r) * T + t0 + tt] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int tt = 0; tt < CW_TC; ++tt) {
            float dv[8], xv[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) dv[i] = dus[(og * 8 + i) * CW_PITCH + tt];
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[q] = xs[(cg + 32 * q) * CW_PITCH + tt];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[i][q] = fmaf(dv[i], xv[q], acc[i][q]);
        }
    }
    float* pb = part + (size_t)blockIdx.x * Co * Ci;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = og * 8 + i, c = cg + 32 * q;
            if (o < Co && c < Ci) pb[(size_t)o * Ci + c] = acc[i][q];
        }
}

// The same weight gradient on the matrix cores (rows of a multiple of four samples): dW = sum over slices of du_slice x_slice^T is a
// 64 x 128 x (B T) GEMM whose two operands are both contiguous along the reduction axis t -- fp32 MFMA (v_mfma_f32_32x32x2_f32:
// exact fp32 products, an fmaf chain per output, no operand splitting and no magnitude bounds).  Wave w owns the 32 input channels
// 32 w .. 32 w + 31 and both 32-row halves of the outputs (2 accumulators); lane (l31, h) reads four consecutive t of its du / x
// row per ds_read_b128 (h selects t .. t+3 or t+4 .. t+7: the MFMA pairs k = t+e of the h = 0 lanes with k = t+4+e of the h = 1
// lanes -- any pairing is a valid reduction order as long as both operands use the same one), i.e. 8 MFMAs per 3 LDS reads.  Row
// pitch 68 floats: the 32 rows of a b128 read spread over all banks.  Same slices, same partials, same fixed-order reduction as
// the VALU kernel above (which at 2.7 FMAs per ds_read_b32 ran at 27 TFLOP/s: 150 us per call; this one ~45).
typedef float cw_f32x16 __attribute__((ext_vector_type(16)));
constexpr int CWM_PITCH = 68;
__global__ void __launch_bounds__(256) chan_contract_bwd_w_mfma_kernel(const float* __restrict__ du, const float* __restrict__ x,
                                                                       float* __restrict__ part, int B, int Ci, int Co, int T) {
    __shared__ __attribute__((aligned(16))) float dus[64 * CWM_PITCH];
    __shared__ __attribute__((aligned(16))) float xs[128 * CWM_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int nchunk = (T + CW_TC - 1) / CW_TC;
    const int nslice = B * nchunk;
    cw_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const bool wave_on = wave * 32 < Ci;                             // block-uniform per wave: its 32 input channels exist
    for (int sl = blockIdx.x; sl < nslice; sl += gridDim.x) {
        const int b = sl / nchunk, t0 = (sl - b * nchunk) * CW_TC;
        // stage: 16 float4 per row, 64 + 128 rows (T % 4 == 0: a float4 is inside the row or past its end)
        float4 v[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int i = tid + 256 * q;                             // 0 .. 3071
            const int r = i >> 4, t4 = (i & 15) * 4;
            const bool isx = r >= 64;
            const int rr = isx ? r - 64 : r;
            const bool ok = (t0 + t4 < T) && (isx ? rr < Ci : rr < Co);
            const float* src = isx ? x + ((size_t)b * Ci + rr) * T : du + ((size_t)b * Co + rr) * T;
            v[q] = ok ? *reinterpret_cast<const float4*>(src + t0 + t4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();                                             // the previous slice has been consumed
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int i = tid + 256 * q;
            const int r = i >> 4, t4 = (i & 15) * 4;
            float* dst = (r >= 64) ? xs + (r - 64) * CWM_PITCH + t4 : dus + r * CWM_PITCH + t4;
            *reinterpret_cast<float4*>(dst) = v[q];
        }
        __syncthreads();
        if (wave_on) {
            const float* a0p = dus + l31 * CWM_PITCH + 4 * h;
            const float* a1p = dus + (32 + l31) * CWM_PITCH + 4 * h;
            const float* bp = xs + (wave * 32 + l31) * CWM_PITCH + 4 * h;
#pragma unroll
            for (int kk = 0; kk < CW_TC; kk += 8) {
                const float4 a0 = *reinterpret_cast<const float4*>(a0p + kk);
                const float4 a1 = *reinterpret_cast<const float4*>(a1p + kk);
                const float4 bq = *reinterpret_cast<const float4*>(bp + kk);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bq.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bq.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bq.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bq.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bq.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bq.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bq.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bq.w, acc[1], 0, 0, 0);
            }
        }
    }
    // accumulator (i, r) of lane (l31, h): output row o = 32 i + 8 (r / 4) + 4 h + r % 4, input channel c = 32 wave + l31
    float* pb = part + (size_t)blockIdx.x * Co * Ci;
    const int c = wave * 32 + l31;
    if (c < Ci) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * i + 8 * (r >> 2) + 4 * h + (r & 3);
                if (o < Co) pb[(size_t)o * Ci + c] = acc[i][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------ BatchNorm + ELU + AvgPool
// block reduction of two doubles (fixed order: lanes by shuffle tree, then waves ascending)
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    if (lane == 0) { sm[2 * wave] = a; sm[2 * wave + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0.0, sb = 0.0;
        for (int w = 0; w < nw; ++w) { sa += sm[2 * w]; sb += sm[2 * w + 1]; }
        sm[0] = sa; sm[1] = sb;
    }
    __syncthreads();
    a = sm[0]; b = sm[1];
}

// per-channel sum v and sum v^2 over the rows b = blockIdx.y, blockIdx.y + gridDim.y, ...: part[(slice, c, 2)] doubles
__global__ void __launch_bounds__(256) chan_stats_kernel(const float* __restrict__ v, double* __restrict__ part, int B, int Cc,
                                                         int T) {
    __shared__ double sm[8];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float* row = v + ((size_t)b * Cc + c) * T;
        float fs = 0.f, fq = 0.f;                           // <= ceil(T/256) values per thread and row in float, rows in double
        for (int t = threadIdx.x; t < T; t += 256) { const float a = row[t]; fs += a; fq = fmaf(a, a, fq); }
        s += (double)fs; q += (double)fq;
    }
    block_sum2(s, q, sm);
    if (threadIdx.x == 0) { part[((size_t)blockIdx.y * Cc + c) * 2] = s; part[((size_t)blockIdx.y * Cc + c) * 2 + 1] = q; }
}

// out[c, 0..1] = sum over slices (ascending) of part
__global__ void chan_stats_finalize_kernel(const double* __restrict__ part, double* __restrict__ out, int nsl, int n2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    double s = 0.0;
    for (int p = 0; p < nsl; ++p) s += part[(size_t)p * n2 + i];
    out[i] = s;
}

__device__ __forceinline__ float elu1(float z) { return z > 0.f ? z : expm1f(z); }        // torch's ELU (alpha 1) uses expm1 too
__device__ __forceinline__ float elu1_grad(float z) { return z > 0.f ? 1.f : __expf(z); }

// out[b,c,tp] = (1/P) sum_{i<P} ELU(s[c] v[b,c,tp P + i] + t[c]),  Tp = T / P (AvgPool drops the remainder)
__global__ void __launch_bounds__(256) affine_elu_pool_kernel(const float* __restrict__ v, const float* __restrict__ sc,
                                                              const float* __restrict__ sh, float* __restrict__ out, int Cc,
                                                              int T, int P, int Tp) {
    const int r = blockIdx.x;                                // row = b*Cc + c
    const int tp = blockIdx.y * 256 + threadIdx.x;
    if (tp >= Tp) return;
    const float s = sc[r % Cc], t = sh[r % Cc];
    const float* row = v + (size_t)r * T + (size_t)tp * P;
    float a = 0.f;
    for (int i = 0; i < P; ++i) a += elu1(fmaf(s, row[i], t));
    out[(size_t)r * Tp + tp] = a / (float)P;
}

// backward sums: S1[c] = sum dz, S2[c] = sum dz * vc, vc = v - m[c];  dz = ELU'(s v + t) * dout[b,c,t/P] / P  (0 past Tp*P)
__global__ void __launch_bounds__(256) bn_elu_pool_bwd_sums_kernel(const float* __restrict__ v, const float* __restrict__ dout,
                                                                   const float* __restrict__ sc, const float* __restrict__ sh,
                                                                   const float* __restrict__ mean, double* __restrict__ part,
                                                                   int B, int Cc, int T, int P, int Tp) {
    __shared__ double sm[8];
    const int c = blockIdx.x;
    const float s = sc[c], t = sh[c], m = mean[c], invP = 1.f / (float)P;
    double s1 = 0.0, s2 = 0.0;
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float* row = v + ((size_t)b * Cc + c) * T;
        const float* drow = dout + ((size_t)b * Cc + c) * Tp;
        float f1 = 0.f, f2 = 0.f;
        for (int tt = threadIdx.x; tt < Tp * P; tt += 256) {
            const float a = row[tt];
            const float dz = elu1_grad(fmaf(s, a, t)) * drow[tt / P] * invP;
            f1 += dz; f2 = fmaf(dz, a - m, f2);
        }
        s1 += (double)f1; s2 += (double)f2;
    }
    block_sum2(s1, s2, sm);
    if (threadIdx.x == 0) { part[((size_t)blockIdx.y * Cc + c) * 2] = s1; part[((size_t)blockIdx.y * Cc + c) * 2 + 1] = s2; }
}

// dv[b,c,t] = ka[c] * dz + kb[c] + kc[c] * v[b,c,t]      (BatchNorm backward folded into three per-channel coefficients)
__global__ void __launch_bounds__(256) bn_elu_pool_bwd_apply_kernel(const float* __restrict__ v, const float* __restrict__ dout,
                                                                    const float* __restrict__ sc, const float* __restrict__ sh,
                                                                    const float* __restrict__ ka, const float* __restrict__ kb,
                                                                    const float* __restrict__ kc, float* __restrict__ dv, int Cc,
                                                                    int T, int P, int Tp) {
    const int r = blockIdx.x, c = r % Cc;
    const int tt = blockIdx.y * 256 + threadIdx.x;
    if (tt >= T) return;
    const float a = v[(size_t)r * T + tt];
    float dz = 0.f;
    if (tt < Tp * P) dz = elu1_grad(fmaf(sc[c], a, sh[c])) * dout[(size_t)r * Tp + tt / P] / (float)P;
    dv[(size_t)r * T + tt] = fmaf(ka[c], dz, fmaf(kc[c], a, kb[c]));
}

// The per-channel algebra between the passes, as ONE tiny launch each way instead of ~25 float64 torch element-wise kernels:
// forward: moments -> (scale, shift) of the apply pass, the saved (mean_v, r) and the running-statistics update;
// backward: (S1, S2) -> the three coefficients of the apply pass and the parameter gradients.   y = alpha v + c.
__global__ void bn_fold_fwd_kernel(const double* __restrict__ sums, const float* __restrict__ alpha, const float* __restrict__ cshift,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ scale,
                                   float* __restrict__ shift, double* __restrict__ fold /* (C,2): mean_v, r */,
                                   float* __restrict__ run_mean, float* __restrict__ run_var, int C, double n, double eps,
                                   double momentum) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[2 * c] / n;
    double var = sums[2 * c + 1] / n - m * m;
    var = var > 0.0 ? var : 0.0;
    const double a = alpha ? (double)alpha[c] : 1.0;
    const double r = 1.0 / sqrt(a * a * var + eps);
    const double sc = (double)gamma[c] * a * r;
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - sc * m);
    fold[2 * c] = m;
    fold[2 * c + 1] = r;
    if (run_mean) {
        const double mean_y = a * m + (cshift ? (double)cshift[c] : 0.0);
        const double var_y = a * a * var * (n > 1.0 ? n / (n - 1.0) : 1.0);           // running_var holds the unbiased estimate
        run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * mean_y);
        run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * var_y);
    }
}

__global__ void bn_fold_bwd_kernel(const double* __restrict__ sums /* (C,2): S1, S2 */, const double* __restrict__ fold,
                                   const float* __restrict__ alpha, const float* __restrict__ gamma, float* __restrict__ ka,
                                   float* __restrict__ kb, float* __restrict__ kc, float* __restrict__ dgamma,
                                   float* __restrict__ dbeta, float* __restrict__ dalpha, int C, double n, double eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double S1 = sums[2 * c], S2 = sums[2 * c + 1];
    const double m = fold[2 * c], r = fold[2 * c + 1];
    const double a = alpha ? (double)alpha[c] : 1.0, g = (double)gamma[c];
    const double k0 = g * a * r;
    const double kcc = -k0 * a * a * r * r * S2 / n;
    ka[c] = (float)k0;
    kc[c] = (float)kcc;
    kb[c] = (float)(-k0 * S1 / n - kcc * m);
    dgamma[c] = (float)(a * r * S2);
    dbeta[c] = (float)S1;
    if (dalpha) dalpha[c] = (float)(g * r * r * r * eps * S2);
}

// ------------------------------------------------------------------------------------------------ C ABI
static int cc_check(const char* who, const void* a, const void* b, const void* c, int B, int Ci, int Co, int T) {
    if (!a || !b || !c || B <= 0 || Ci <= 0 || Co <= 0 || T <= 0) {
        ign_set_error("%s: null pointer or bad dimension (B=%d Ci=%d Co=%d T=%d)", who, B, Ci, Co, T);
        return IGN_E_ARG;
    }
    if (Ci > 128 || Co > 64 || B > 65535) {
        ign_set_error("%s: Ci=%d (<= 128), Co=%d (<= 64) or B=%d (<= 65535) outside the kernel's tile", who, Ci, Co, B);
        return IGN_E_UNSUP;
    }
    return 0;
}

extern "C" int ign_chan_contract_fwd(const float* x_bct, const float* wt_ci64, float* u_bot, int B, int Ci, int Co, int T,
                                     void* stream) {
    int rc;
    if ((rc = cc_check("ign_chan_contract_fwd", x_bct, wt_ci64, u_bot, B, Ci, Co, T))) return rc;
    hipStream_t s = (hipStream_t)stream;
    IgnScopedTimer tm("chan_contract", s);
    hipLaunchKernelGGL(chan_contract_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)B), dim3(256), 0, s, x_bct, wt_ci64, u_bot,
                       Ci, Co, T);
    return ign_check_launch("chan_contract_kernel");
}

static int cw_blocks(int B, int T) { return std::max(1, std::min(512, B * ((T + CW_TC - 1) / CW_TC))); }

extern "C" size_t ign_chan_contract_bwd_weight_workspace_bytes(int B, int Ci, int Co, int T) {
    if (B <= 0 || Ci <= 0 || Co <= 0 || T <= 0) return 0;
    return (size_t)cw_blocks(B, T) * Co * Ci * sizeof(float);
}

extern "C" int ign_chan_contract_bwd_weight(const float* du_bot, const float* x_bct, float* dw_oc, void* workspace, int B, int Ci,
                                            int Co, int T, void* stream) {
    int rc;
    if ((rc = cc_check("ign_chan_contract_bwd_weight", du_bot, x_bct, dw_oc, B, Ci, Co, T))) return rc;
    if (!workspace) { ign_set_error("ign_chan_contract_bwd_weight: null workspace"); return IGN_E_ARG; }
    hipStream_t s = (hipStream_t)stream;
    const int nb = cw_blocks(B, T);
    {
        IgnScopedTimer tm("chan_contract_bwd_w", s);
        if (T % 4 == 0 && !((uintptr_t)du_bot & 15) && !((uintptr_t)x_bct & 15))
            hipLaunchKernelGGL(chan_contract_bwd_w_mfma_kernel, dim3(nb), dim3(256), 0, s, du_bot, x_bct, (float*)workspace, B, Ci, Co, T);
        else
            hipLaunchKernelGGL(chan_contract_bwd_w_kernel, dim3(nb), dim3(256), 0, s, du_bot, x_bct, (float*)workspace, B, Ci, Co, T);
    }
    if ((rc = ign_check_launch("chan_contract_bwd_w_kernel"))) return rc;
    ign_launch_reduce_parts((const float*)workspace, dw_oc, nb, (size_t)Co * Ci, s);
    return ign_check_launch("reduce_parts_kernel");
}

static int stat_slices(int B) { return std::max(1, std::min(B, 32)); }

extern "C" size_t ign_chan_stats_workspace_bytes(int B, int Cc) {
    if (B <= 0 || Cc <= 0) return 0;
    return (size_t)stat_slices(B) * Cc * 2 * sizeof(double);
}

static int bn_check(const char* who, int B, int Cc, int T, int P) {
    if (B <= 0 || Cc <= 0 || T <= 0 || P <= 0 || P > T || Cc > 65535 || (long long)B * Cc > 2147483647LL || T > 65535 * 256) {
        ign_set_error("%s: bad dimension (B=%d C=%d T=%d P=%d)", who, B, Cc, T, P);
        return IGN_E_ARG;
    }
    return 0;
}

// sums_c2 (Cc, 2) doubles: sum v, sum v^2 over (b, t)
extern "C" int ign_chan_stats(const float* v_bct, double* sums_c2, void* workspace, int B, int Cc, int T, void* stream) {
    int rc;
    if (!v_bct || !sums_c2 || !workspace) { ign_set_error("ign_chan_stats: null pointer"); return IGN_E_ARG; }
    if ((rc = bn_check("ign_chan_stats", B, Cc, T, 1))) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int nsl = stat_slices(B);
    IgnScopedTimer tm("chan_stats", s);
    hipLaunchKernelGGL(chan_stats_kernel, dim3((unsigned)Cc, (unsigned)nsl), dim3(256), 0, s, v_bct, (double*)workspace, B, Cc, T);
    if ((rc = ign_check_launch("chan_stats_kernel"))) return rc;
    hipLaunchKernelGGL(chan_stats_finalize_kernel, dim3((unsigned)((2 * Cc + 127) / 128)), dim3(128), 0, s, (const double*)workspace,
                       sums_c2, nsl, 2 * Cc);
    return ign_check_launch("chan_stats_finalize_kernel");
}

extern "C" int ign_affine_elu_pool_fwd(const float* v_bct, const float* scale_c, const float* shift_c, float* out, int B, int Cc,
                                       int T, int P, void* stream) {
    int rc;
    if (!v_bct || !scale_c || !shift_c || !out) { ign_set_error("ign_affine_elu_pool_fwd: null pointer"); return IGN_E_ARG; }
    if ((rc = bn_check("ign_affine_elu_pool_fwd", B, Cc, T, P))) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int Tp = T / P;
    IgnScopedTimer tm("affine_elu_pool", s);
    hipLaunchKernelGGL(affine_elu_pool_kernel, dim3((unsigned)(B * Cc), (unsigned)((Tp + 255) / 256)), dim3(256), 0, s, v_bct, scale_c,
                       shift_c, out, Cc, T, P, Tp);
    return ign_check_launch("affine_elu_pool_kernel");
}

// sums_c2 (Cc, 2) doubles: S1 = sum dz, S2 = sum dz (v - mean)
extern "C" int ign_bn_elu_pool_bwd_sums(const float* v_bct, const float* dout, const float* scale_c, const float* shift_c,
                                        const float* mean_c, double* sums_c2, void* workspace, int B, int Cc, int T, int P,
                                        void* stream) {
    int rc;
    if (!v_bct || !dout || !scale_c || !shift_c || !mean_c || !sums_c2 || !workspace) {
        ign_set_error("ign_bn_elu_pool_bwd_sums: null pointer");
        return IGN_E_ARG;
    }
    if ((rc = bn_check("ign_bn_elu_pool_bwd_sums", B, Cc, T, P))) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int nsl = stat_slices(B);
    IgnScopedTimer tm("bn_elu_pool_bwd_sums", s);
    hipLaunchKernelGGL(bn_elu_pool_bwd_sums_kernel, dim3((unsigned)Cc, (unsigned)nsl), dim3(256), 0, s, v_bct, dout, scale_c, shift_c,
                       mean_c, (double*)workspace, B, Cc, T, P, T / P);
    if ((rc = ign_check_launch("bn_elu_pool_bwd_sums_kernel"))) return rc;
    hipLaunchKernelGGL(chan_stats_finalize_kernel, dim3((unsigned)((2 * Cc + 127) / 128)), dim3(128), 0, s, (const double*)workspace,
                       sums_c2, nsl, 2 * Cc);
    return ign_check_launch("chan_stats_finalize_kernel");
}

extern "C" int ign_bn_elu_pool_bwd_apply(const float* v_bct, const float* dout, const float* scale_c, const float* shift_c,
                                         const float* ka_c, const float* kb_c, const float* kc_c, float* dv_bct, int B, int Cc,
                                         int T, int P, void* stream) {
    int rc;
    if (!v_bct || !dout || !scale_c || !shift_c || !ka_c || !kb_c || !kc_c || !dv_bct) {
        ign_set_error("ign_bn_elu_pool_bwd_apply: null pointer");
        return IGN_E_ARG;
    }
    if ((rc = bn_check("ign_bn_elu_pool_bwd_apply", B, Cc, T, P))) return rc;
    hipStream_t s = (hipStream_t)stream;
    IgnScopedTimer tm("bn_elu_pool_bwd_apply", s);
    hipLaunchKernelGGL(bn_elu_pool_bwd_apply_kernel, dim3((unsigned)(B * Cc), (unsigned)((T + 255) / 256)), dim3(256), 0, s, v_bct, dout,
                       scale_c, shift_c, ka_c, kb_c, kc_c, dv_bct, Cc, T, P, T / P);
    return ign_check_launch("bn_elu_pool_bwd_apply_kernel");
}

extern "C" int ign_bn_fold_fwd(const double* sums_c2, const float* alpha_c, const float* cshift_c, const float* gamma_c,
                               const float* beta_c, float* scale_c, float* shift_c, double* fold_c2, float* running_mean,
                               float* running_var, int C, long long n, float eps, float momentum, void* stream) {
    if (!sums_c2 || !gamma_c || !beta_c || !scale_c || !shift_c || !fold_c2 || C <= 0 || n <= 0 || (!running_mean) != (!running_var)) {
        ign_set_error("ign_bn_fold_fwd: null pointer or bad dimension (C=%d n=%lld)", C, n);
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(bn_fold_fwd_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, sums_c2, alpha_c, cshift_c,
                       gamma_c, beta_c, scale_c, shift_c, fold_c2, running_mean, running_var, C, (double)n, (double)eps, (double)momentum);
    return ign_check_launch("bn_fold_fwd_kernel");
}

extern "C" int ign_bn_fold_bwd(const double* sums_c2, const double* fold_c2, const float* alpha_c, const float* gamma_c, float* ka_c,
                               float* kb_c, float* kc_c, float* dgamma_c, float* dbeta_c, float* dalpha_c, int C, long long n, float eps,
                               void* stream) {
    if (!sums_c2 || !fold_c2 || !gamma_c || !ka_c || !kb_c || !kc_c || !dgamma_c || !dbeta_c || C <= 0 || n <= 0 ||
        (!alpha_c) != (!dalpha_c)) {
        ign_set_error("ign_bn_fold_bwd: null pointer or bad dimension (C=%d n=%lld)", C, n);
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, sums_c2, fold_c2, alpha_c,
                       gamma_c, ka_c, kb_c, kc_c, dgamma_c, dbeta_c, dalpha_c, C, (double)n, (double)eps);
    return ign_check_launch("bn_fold_bwd_kernel");
}

// BatchNorm1d / ReLU / average-pool glue of the FCN expert (IGN/model/FullyConvNet.py:33-34,40-41,47-49,55-57) around the
// convolution GEMMs of ign_clconv_*.hip: partial-sum finalisation, the last block's BatchNorm+ReLU+pool and its backward,
// and the elementwise BatchNorm-backward pass that writes dL/dy in the zero-padded layout the GEMMs read.
#include "ign_clconv.h"

// maxima of non-negative floats as unsigned integers on their bit patterns (see 'operand bounds' below)
__device__ __forceinline__ void atomic_absmax(float* slot, float v) { ign_atomic_absmax(slot, v); }      // v >= 0
__device__ __forceinline__ float block_max_1024(float v, float* sh) {                // result valid in thread 0
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 1; i < nw; ++i) v = fmaxf(v, sh[i]);
    }
    return v;
}


// ------------------------------------------------------------------------------------------------ BatchNorm glue
// Sum of the per-tile partials (nparts, 2, C) for BN_CH channels per block: BN_SL slices of the partials are summed in
// parallel (ascending inside a slice, fp32 over <= 64 addends, then double), and the slice sums are combined in fixed order
// in double.  Small blocks of channels keep C/BN_CH workgroups busy on a reduction that is only a few MB.
constexpr int BN_CH = 8, BN_SL = 128;
__device__ __forceinline__ void bn_sum_partials(const float* __restrict__ part, int nparts, int C, double* s_out, double* q_out) {
    __shared__ double sh[2][BN_SL][BN_CH + 1];
    const int cl = threadIdx.x % BN_CH, sl = threadIdx.x / BN_CH;
    const int c = blockIdx.x * BN_CH + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        const int per = (nparts + BN_SL - 1) / BN_SL;
        const int i0 = sl * per, i1 = min(nparts, i0 + per);
        float fs = 0.f, fq = 0.f;
        int n = 0;
        for (int i = i0; i < i1; ++i) {
            fs += part[((size_t)i * 2 + 0) * C + c];
            fq += part[((size_t)i * 2 + 1) * C + c];
            if (++n == 64) { s += (double)fs; q += (double)fq; fs = fq = 0.f; n = 0; }
        }
        s += (double)fs; q += (double)fq;
    }
    sh[0][sl][cl] = s;
    sh[1][sl][cl] = q;
    __syncthreads();
    s = q = 0.0;
    if (sl == 0) {
        for (int k = 0; k < BN_SL; ++k) { s += sh[0][k][cl]; q += sh[1][k][cl]; }
    }
    *s_out = s; *q_out = q;
}

// forward finalize: mean, biased var -> a = gamma*invstd, b = beta - a*mean; running stats (unbiased var).
__global__ void __launch_bounds__(1024) bn_finalize_fwd_kernel(const float* __restrict__ part, int nparts, long long R, int C,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                       float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ a,
                                       float* __restrict__ b, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    double s, q;
    bn_sum_partials(part, nparts, C, &s, &q);
    const int c = blockIdx.x * BN_CH + threadIdx.x % BN_CH;
    if (threadIdx.x / BN_CH != 0 || c >= C) return;
    const double m = s / (double)R;
    double var = q / (double)R - m * m;
    if (var < 0.0) var = 0.0;
    const float inv = (float)(1.0 / sqrt(var + (double)eps));
    const float av = gamma[c] * inv;
    a[c] = av;
    b[c] = beta[c] - av * (float)m;
    mean_out[c] = (float)m;
    invstd_out[c] = inv;
    if (run_mean) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)m;
        const double unb = (R > 1) ? var * (double)R / (double)(R - 1) : var;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
    }
}

__global__ void bn_affine_eval_kernel(const float* __restrict__ run_mean, const float* __restrict__ run_var,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int C,
                                      float* __restrict__ a, float* __restrict__ b, float* __restrict__ mean_out,
                                      float* __restrict__ invstd_out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float inv = 1.f / sqrtf(run_var[c] + eps);
    const float av = gamma[c] * inv;
    a[c] = av;
    b[c] = beta[c] - av * run_mean[c];
    mean_out[c] = run_mean[c];
    invstd_out[c] = inv;
}

__global__ void __launch_bounds__(1024) bn_finalize_bwd_kernel(const float* __restrict__ part, int nparts, int C,
                                                               float* __restrict__ dbeta, float* __restrict__ dgamma) {
    double s, q;
    bn_sum_partials(part, nparts, C, &s, &q);
    const int c = blockIdx.x * BN_CH + threadIdx.x % BN_CH;
    if (threadIdx.x / BN_CH != 0 || c >= C) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
}

// Global average pool of relu(a*y + b) over time:  pooled[b][c] = (1/T) sum_t max(a_c*y[b,t,c] + b_c, 0).
// One block per sample; thread <-> (row lane, 4 channels); the row lanes are combined through LDS in fixed order.
// With a head (W != null): logits[b][n] = bias[n] + sum_c pooled[b][c] W[n][c] from the same block -- the sample's pooled row is
// complete here (one block per sample), so the class head of the FCN expert (IGN/model/FullyConvNet.py:58) needs no launch.
__global__ void __launch_bounds__(256) bn_relu_pool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ a,
                                                               const float* __restrict__ b, float* __restrict__ pooled, int T,
                                                               int C, const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ logits, int N) {
    extern __shared__ float sm[];                         // [rif][C]
    const int c4 = C / 4, rif = 256 / c4;
    const int col = (threadIdx.x % c4) * 4, rofs = threadIdx.x / c4;
    const float* yb = y + (size_t)blockIdx.x * T * C;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (rofs < rif) {
        const float4 av = *reinterpret_cast<const float4*>(a + col), bv = *reinterpret_cast<const float4*>(b + col);
        for (int t = rofs; t < T; t += rif) {
            const float4 yv = *reinterpret_cast<const float4*>(yb + (size_t)t * C + col);
            s[0] += fmaxf(fmaf(av.x, yv.x, bv.x), 0.f);
            s[1] += fmaxf(fmaf(av.y, yv.y, bv.y), 0.f);
            s[2] += fmaxf(fmaf(av.z, yv.z, bv.z), 0.f);
            s[3] += fmaxf(fmaf(av.w, yv.w, bv.w), 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) sm[rofs * C + col + q] = s[q];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float u = 0.f;
        for (int q = 0; q < rif; ++q) u += sm[q * C + c];
        u /= (float)T;
        pooled[(size_t)blockIdx.x * C + c] = u;
        if (W) sm[c] = u;                                 // column c is this thread's alone: read above, rewritten here
    }
    if (!W) return;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int n = wave; n < N; n += 4) {                   // one wave per class: fixed-order lane partials, butterfly sum
        float u = 0.f;
        for (int c = lane; c < C; c += 64) u = fmaf(sm[c], W[(size_t)n * C + c], u);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) u += __shfl_xor(u, o, 64);
        if (lane == 0) logits[(size_t)blockIdx.x * N + n] = u + (bias ? bias[n] : 0.f);
    }
}

// Backward of the pool through ReLU: g[b,t,c] = (gpool[b,c]/T) * [a*y + b > 0]; per-block partials of sum g and
// sum g*yhat (BatchNorm backward).  Grid (ceil(T / rows_per_block), B).
constexpr int POOL_ROWS = 128;
__global__ void __launch_bounds__(256) bn_relu_pool_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gpool,
                                                               const float* __restrict__ a, const float* __restrict__ b,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               float* __restrict__ g, float* __restrict__ part, int T, int C) {
    extern __shared__ float sm[];                         // [2][rif][C]
    const int c4 = C / 4, rif = 256 / c4;
    const int col = (threadIdx.x % c4) * 4, rofs = threadIdx.x / c4;
    const int bi = blockIdx.y;
    const int t0 = blockIdx.x * POOL_ROWS, t1 = min(T, t0 + POOL_ROWS);
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (rofs < rif) {
        float av[4], bv[4], mv[4], iv[4], gp[4];
        vload<4>(av, a + col); vload<4>(bv, b + col); vload<4>(mv, mean + col); vload<4>(iv, invstd + col);
        vload<4>(gp, gpool + (size_t)bi * C + col);
        const float invT = 1.f / (float)T;
        for (int t = t0 + rofs; t < t1; t += rif) {
            const size_t off = ((size_t)bi * T + t) * C + col;
            float yy[4], gg[4];
            vload<4>(yy, y + off);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gg[q] = (fmaf(av[q], yy[q], bv[q]) > 0.f) ? gp[q] * invT : 0.f;
                s0[q] += gg[q];
                s1[q] = fmaf(gg[q], (yy[q] - mv[q]) * iv[q], s1[q]);
            }
            vstore<4>(g + off, gg);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { sm[rofs * C + col + q] = s0[q]; sm[(rif + rofs) * C + col + q] = s1[q]; }
    }
    __syncthreads();
    const size_t pi = (size_t)bi * gridDim.x + blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float u = 0.f, v = 0.f;
        for (int q = 0; q < rif; ++q) { u += sm[q * C + c]; v += sm[(rif + q) * C + c]; }
        part[(pi * 2 + 0) * C + c] = u;
        part[(pi * 2 + 1) * C + c] = v;
    }
}

// dy = a * (g - [training] (dbeta + yhat*dgamma)/R) = a*g + c1*y + c0 written into a per-sample zero-padded buffer
// (B, pad + T + pad, C); the pad rows are zeroed here.  Thread <-> (row lane, 4 channels): the three per-channel
// coefficients are formed once per thread, then the block walks APPLY_ROWS padded rows.
constexpr int APPLY_ROWS = 64;
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                           const float* __restrict__ a, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ dbeta,
                                                           const float* __restrict__ dgamma, float* __restrict__ dyp, int T, int C,
                                                           int pad, long long rows_padded, float invR, int training,
                                                           float* __restrict__ amax_slot) {
    const int c4 = C / 4, rif = 256 / c4;
    const int col = (threadIdx.x % c4) * 4, rofs = threadIdx.x / c4;
    if (rofs >= rif) return;
    float amax = 0.f;
    float ca[4], c1[4], c0[4];
    vload<4>(ca, a + col);
    if (training) {
        float mv[4], iv[4], db[4], dg[4];
        vload<4>(mv, mean + col); vload<4>(iv, invstd + col); vload<4>(db, dbeta + col); vload<4>(dg, dgamma + col);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c1[q] = -ca[q] * iv[q] * dg[q] * invR;
            c0[q] = -ca[q] * (db[q] - mv[q] * iv[q] * dg[q]) * invR;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) c1[q] = c0[q] = 0.f;
    }
    const int Tp = T + 2 * pad;
    const long long r0 = (long long)blockIdx.x * APPLY_ROWS;
    const long long r1 = min(rows_padded, r0 + APPLY_ROWS);
    for (long long row = r0 + rofs; row < r1; row += rif) {
        const long long bi = row / Tp;
        const int tp = (int)(row - bi * Tp);
        float out[4] = {0.f, 0.f, 0.f, 0.f};
        if (tp >= pad && tp < pad + T) {
            const size_t off = ((size_t)bi * T + (tp - pad)) * C + col;
            float gg[4], yy[4] = {0.f, 0.f, 0.f, 0.f};
            vload<4>(gg, g + off);
            if (training) vload<4>(yy, y + off);
#pragma unroll
            for (int q = 0; q < 4; ++q) { out[q] = fmaf(ca[q], gg[q], fmaf(c1[q], yy[q], c0[q])); amax = fmaxf(amax, fabsf(out[q])); }
        }
        vstore<4>(dyp + (size_t)row * C + col, out);
    }
    if (amax_slot) {              // max |dL/dy| for the fp16 GEMMs that read this tensor (bitwise reproducible: integer max)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if ((threadIdx.x & 63) == 0) atomic_absmax(amax_slot, amax);
    }
}

// ------------------------------------------------------------------------------------------------ operand bounds (fp16 path)
// The two-plane fp16 GEMMs (ign_clconv_*_h3) scale each operand tensor by a power of two taken from an upper bound of its
// magnitude (ign_pow2_scale).  Bounds live in a small device array of `slots` (4 floats per layer):
//   slot 4l + 0  max |W_l|                                    (fcn_scan_kernel, from the parameters)
//   slot 4l + 1  bound of the layer's INPUT as the GEMM sees it: l = 0: max |x| (absmax_kernel, from the data); l > 0:
//                max_c ( |gamma_c| sqrt(R - 1) + |beta_c| ) of the BatchNorm in front -- a hard bound of relu(gamma yhat + beta),
//                because a value standardised with the batch's own mean and variance over R rows cannot exceed sqrt(R - 1)
//   slot 4l + 2  max |dL/dy_l| (bn_bwd_apply_kernel, as it writes the tensor; zeroed by fcn_scan_kernel at the start of the step)
// Maxima of non-negative floats are taken as unsigned integers on their bit patterns (order-preserving): atomicMax is exact
// and order-independent, so the bounds -- and everything computed from them -- are bitwise reproducible.
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, long long n, float* __restrict__ slot) {
    __shared__ float sh[16];
    float m = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
    m = block_max_1024(m, sh);
    if (threadIdx.x == 0) atomic_absmax(slot, m);
}

// max |x| of up to 16 tensors in one launch (the weights of a model's dense layers: one launch per step instead of one per layer);
// grid.y = tensor, grid.x = chunks of it, atomic maximum into the tensor's own slot (caller zeroes the slots).
constexpr int AMM_MAX = 16;
struct AbsmaxMultiTable { const float* x[AMM_MAX]; long long n[AMM_MAX]; float* slot[AMM_MAX]; };
__global__ void __launch_bounds__(256) absmax_multi_kernel(const AbsmaxMultiTable t) {
    __shared__ float sh[16];
    const int g = blockIdx.y;
    const float* __restrict__ x = t.x[g];
    const long long n = t.n[g];
    float m = 0.f;
    if (((uintptr_t)x & 15) == 0) {
        const long long n4 = n >> 2;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        if (blockIdx.x == 0)
            for (long long i = (n4 << 2) + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    }
    m = block_max_1024(m, sh);
    if (threadIdx.x == 0) atomic_absmax(t.slot[g], m);
}

constexpr int SCAN_LMAX = 8;
struct FcnScanTable {
    const float* w[SCAN_LMAX]; long long nw[SCAN_LMAX];
    const float* gamma[SCAN_LMAX]; const float* beta[SCAN_LMAX]; int C[SCAN_LMAX]; float sqrtR[SCAN_LMAX];
};
__global__ void __launch_bounds__(1024) fcn_scan_kernel(const FcnScanTable t, float* __restrict__ slots, float* __restrict__ zero,
                                                        long long nzero) {
    __shared__ float sh[16];
    const int l = blockIdx.x;
    for (long long i = (long long)l * 1024 + threadIdx.x; i < nzero; i += (long long)gridDim.x * 1024) zero[i] = 0.f;
    float m = 0.f;
    for (long long i = threadIdx.x; i < t.nw[l]; i += 1024) m = fmaxf(m, fabsf(t.w[l][i]));
    m = block_max_1024(m, sh);
    if (threadIdx.x == 0) slots[4 * l + 0] = m;
    __syncthreads();
    float a = 0.f;
    if (t.gamma[l])
        for (int c = threadIdx.x; c < t.C[l]; c += 1024) a = fmaxf(a, fmaf(fabsf(t.gamma[l][c]), t.sqrtR[l], fabsf(t.beta[l][c])));
    a = block_max_1024(a, sh);
    if (threadIdx.x == 0) { slots[4 * l + 1] = a; slots[4 * l + 2] = 0.f; slots[4 * l + 3] = 0.f; }
}

extern "C" int ign_absmax(const float* x, long long n, float* slot, void* stream) {
    if (!x || !slot || n <= 0 || ((uintptr_t)x & 15)) { ign_set_error("ign_absmax: null / unaligned pointer or n <= 0"); return IGN_E_ARG; }
    const long long blocks = (n / 4 + 256 * 8 - 1) / (256 * 8);
    IgnScopedTimer tm("absmax", (hipStream_t)stream);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks))), dim3(256), 0,
                       (hipStream_t)stream, x, n, slot);
    return ign_check_launch("absmax_kernel");
}

extern "C" int ign_absmax_multi(int n, const float* const* x, const long long* count, float* const* slots, void* stream) {
    if (n <= 0 || n > AMM_MAX || !x || !count || !slots) { ign_set_error("ign_absmax_multi: n=%d outside 1..%d or null table", n, AMM_MAX); return IGN_E_ARG; }
    AbsmaxMultiTable t;
    long long nmax = 0;
    for (int i = 0; i < n; ++i) {
        if (!x[i] || !slots[i] || count[i] <= 0) { ign_set_error("ign_absmax_multi: tensor %d: null pointer or empty", i); return IGN_E_ARG; }
        t.x[i] = x[i]; t.n[i] = count[i]; t.slot[i] = slots[i];
        nmax = count[i] > nmax ? count[i] : nmax;
    }
    long long blocks = (nmax / 4 + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
    IgnScopedTimer tm("absmax", (hipStream_t)stream);
    hipLaunchKernelGGL(absmax_multi_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    return ign_check_launch("absmax_multi_kernel");
}

extern "C" int ign_fcn_scan(int nl, const float* const* w, const long long* nw, const float* const* gamma_prev,
                            const float* const* beta_prev, const int* C_prev, const long long* R_prev, float* slots, float* zero,
                            long long nzero, void* stream) {
    if (nzero < 0 || (nzero > 0 && !zero)) { ign_set_error("ign_fcn_scan: nzero=%lld with zero=%p", nzero, (void*)zero); return IGN_E_ARG; }
    if (nl <= 0 || nl > SCAN_LMAX || !w || !nw || !slots) { ign_set_error("ign_fcn_scan: nl=%d outside 1..%d or null table", nl, SCAN_LMAX); return IGN_E_ARG; }
    FcnScanTable t;
    for (int l = 0; l < nl; ++l) {
        if (!w[l] || nw[l] <= 0) { ign_set_error("ign_fcn_scan: layer %d: null weights", l); return IGN_E_ARG; }
        t.w[l] = w[l]; t.nw[l] = nw[l];
        const bool has = gamma_prev && gamma_prev[l];
        if (has && (!beta_prev || !beta_prev[l] || !C_prev || !R_prev || C_prev[l] <= 0 || R_prev[l] <= 1)) {
            ign_set_error("ign_fcn_scan: layer %d: BatchNorm in front needs beta, C > 0 and R > 1", l);
            return IGN_E_ARG;
        }
        t.gamma[l] = has ? gamma_prev[l] : nullptr; t.beta[l] = has ? beta_prev[l] : nullptr;
        t.C[l] = has ? C_prev[l] : 0; t.sqrtR[l] = has ? sqrtf((float)(R_prev[l] - 1)) : 0.f;
    }
    hipLaunchKernelGGL(fcn_scan_kernel, dim3(nl), dim3(1024), 0, (hipStream_t)stream, t, slots, zero, nzero);
    return ign_check_launch("fcn_scan_kernel");
}

// ------------------------------------------------------------------------------------------------ C ABI
static int bn_check(const char* who, long long R, int C) {
    if (R <= 0 || C <= 0 || (C & 3) || C > 1024) {
        ign_set_error("%s: need R > 0 and 4 <= C <= 1024 with C %% 4 == 0 (R=%lld C=%d)", who, R, C);
        return IGN_E_ARG;
    }
    return 0;
}

extern "C" int ign_bn_finalize_fwd(const float* part, int nparts, long long R, int C, const float* gamma, const float* beta,
                                   float eps, float momentum, float* running_mean, float* running_var, float* a, float* b,
                                   float* mean, float* invstd, void* stream) {
    if (!part || nparts <= 0 || R <= 0 || C <= 0 || !gamma || !beta || !a || !b || !mean || !invstd) {
        ign_set_error("ign_bn_finalize_fwd: bad argument");
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + BN_CH - 1) / BN_CH), dim3(BN_CH * BN_SL), 0, (hipStream_t)stream, part, nparts, R, C, gamma, beta,
                       eps, momentum, running_mean, running_var, a, b, mean, invstd);
    return ign_check_launch("bn_finalize_fwd_kernel");
}

extern "C" int ign_bn_affine_eval(const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                                  float eps, int C, float* a, float* b, float* mean, float* invstd, void* stream) {
    if (!running_mean || !running_var || !gamma || !beta || C <= 0 || !a || !b || !mean || !invstd) {
        ign_set_error("ign_bn_affine_eval: bad argument");
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(bn_affine_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, running_mean, running_var,
                       gamma, beta, eps, C, a, b, mean, invstd);
    return ign_check_launch("bn_affine_eval_kernel");
}

extern "C" int ign_bn_finalize_bwd(const float* part, int nparts, int C, float* dbeta, float* dgamma, void* stream) {
    if (!part || nparts <= 0 || C <= 0 || !dbeta || !dgamma) {
        ign_set_error("ign_bn_finalize_bwd: bad argument");
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + BN_CH - 1) / BN_CH), dim3(BN_CH * BN_SL), 0, (hipStream_t)stream, part, nparts, C, dbeta, dgamma);
    return ign_check_launch("bn_finalize_bwd_kernel");
}

extern "C" int ign_bn_relu_pool_fwd(const float* y, const float* a, const float* b, float* pooled, int B, int T, int C,
                                    void* stream) {
    int rc;
    if ((rc = bn_check("ign_bn_relu_pool_fwd", (long long)B * T, C))) return rc;
    if (!y || !a || !b || !pooled) { ign_set_error("ign_bn_relu_pool_fwd: null pointer"); return IGN_E_ARG; }
    const int rif = 256 / (C / 4);
    IgnScopedTimer tm("bn_relu_pool_fwd", (hipStream_t)stream);
    hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(B), dim3(256), (size_t)rif * C * 4, (hipStream_t)stream, y, a, b, pooled, T, C,
                       (const float*)nullptr, (const float*)nullptr, (float*)nullptr, 0);
    return ign_check_launch("bn_relu_pool_fwd_kernel");
}

extern "C" int ign_bn_relu_pool_head_fwd(const float* y, const float* a, const float* b, float* pooled, const float* W,
                                         const float* bias, float* logits, int B, int T, int C, int N, void* stream) {
    int rc;
    if ((rc = bn_check("ign_bn_relu_pool_head_fwd", (long long)B * T, C))) return rc;
    if (!y || !a || !b || !pooled || !W || !logits || N <= 0) {
        ign_set_error("ign_bn_relu_pool_head_fwd: null pointer or N=%d", N);
        return IGN_E_ARG;
    }
    const int rif = 256 / (C / 4);
    IgnScopedTimer tm("bn_relu_pool_fwd", (hipStream_t)stream);
    hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(B), dim3(256), (size_t)rif * C * 4, (hipStream_t)stream, y, a, b, pooled, T, C,
                       W, bias, logits, N);
    return ign_check_launch("bn_relu_pool_fwd_kernel");
}

extern "C" long long ign_bn_relu_pool_bwd_parts(int B, int T) { return (long long)B * ((T + POOL_ROWS - 1) / POOL_ROWS); }

extern "C" int ign_bn_relu_pool_bwd(const float* y, const float* gpool, const float* a, const float* b, const float* mean,
                                    const float* invstd, float* g, float* part, int B, int T, int C, void* stream) {
    int rc;
    if ((rc = bn_check("ign_bn_relu_pool_bwd", (long long)B * T, C))) return rc;
    if (!y || !gpool || !a || !b || !mean || !invstd || !g || !part) { ign_set_error("ign_bn_relu_pool_bwd: null pointer"); return IGN_E_ARG; }
    const int rif = 256 / (C / 4);
    IgnScopedTimer tm("bn_relu_pool_bwd", (hipStream_t)stream);
    hipLaunchKernelGGL(bn_relu_pool_bwd_kernel, dim3((T + POOL_ROWS - 1) / POOL_ROWS, B), dim3(256), (size_t)2 * rif * C * 4,
                       (hipStream_t)stream, y, gpool, a, b, mean, invstd, g, part, T, C);
    return ign_check_launch("bn_relu_pool_bwd_kernel");
}

extern "C" int ign_bn_bwd_apply(const float* g, const float* y, const float* a, const float* mean, const float* invstd,
                                const float* dbeta, const float* dgamma, float* dyp, int B, int T, int C, int pad, int training,
                                void* stream) {
    return ign_bn_bwd_apply_amax(g, y, a, mean, invstd, dbeta, dgamma, dyp, nullptr, B, T, C, pad, training, stream);
}

extern "C" int ign_bn_bwd_apply_amax(const float* g, const float* y, const float* a, const float* mean, const float* invstd,
                                     const float* dbeta, const float* dgamma, float* dyp, float* amax_slot, int B, int T, int C,
                                     int pad, int training, void* stream) {
    int rc;
    if ((rc = bn_check("ign_bn_bwd_apply", (long long)B * T, C))) return rc;
    if (!g || !a || !dyp || pad < 0 || (training && (!y || !mean || !invstd || !dbeta || !dgamma))) {
        ign_set_error("ign_bn_bwd_apply: null pointer / negative pad");
        return IGN_E_ARG;
    }
    const long long rows = (long long)B * (T + 2 * pad);
    IgnScopedTimer tm("bn_bwd_apply", (hipStream_t)stream);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((rows + APPLY_ROWS - 1) / APPLY_ROWS)), dim3(256), 0, (hipStream_t)stream,
                       g, y, a, mean, invstd, dbeta, dgamma, dyp, T, C, pad, rows, 1.0f / (float)((long long)B * T), training, amax_slot);
    return ign_check_launch("bn_bwd_apply_kernel");
}


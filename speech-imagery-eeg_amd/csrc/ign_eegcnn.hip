// EEG-CNN convolution block (IGN/model/eegcnn.py:67-108) without its 1 GB intermediates.
//
// Block 1 of the reference is  temporal conv (1 -> F1 filters, 1 x k1, 'same', no bias) -> BatchNorm2d(F1) ->
// depthwise spatial conv (C x 1, groups F1, D outputs per group, no bias) -> BatchNorm2d(F1*D) -> ELU -> AvgPool.
// The temporal conv has ONE input channel and the spatial conv only mixes the electrode axis, and BatchNorm is an
// affine map per filter, so with  u[b,o,t] = sum_c w2[o,c] x[b,c,t]  (a 64 x 122 GEMM per time step)
//     y2[b,o,t] = a_f * (w1[f] (*) u[b,o,:])[t] + b_f * sum_c w2[o,c],      f = o / D,
// where a_f, b_f fold BatchNorm-1.  What cannot be moved behind the channel contraction are BatchNorm-1's BATCH
// STATISTICS: mean and variance of y1[b,f,c,t] = (w1[f] (*) x[b,c,:])[t] over (b,c,t).  The mean is a closed form of
// shifted sums of x (host side, autograd); the variance needs  sum (y1 - mu)^2  -- this file computes it, and its
// gradient w.r.t. w1, by brute force on the VALU *without ever storing y1* (B*F1*C*T floats = 1 GB at B=256):
//   ign_conv1_sumsq_fwd : M2[f] = sum_{rows,t} (y1 - mu_f)^2                      B*C*T*F1*k1 = 3.1e10 FMA
//   ign_conv1_sumsq_bwd : G[f,j] = sum_{rows,t} (y1 - mu_f) * xpad[t + j]          2x that (recompute y1, correlate)
// plus the depthwise temporal convolution used for  w1[f] (*) u  and for block 2 (IGN/model/eegcnn.py:78):
//   ign_dwconv1d_fwd / _bwd_data / _bwd_weight : per-channel 1-D 'same' convolution, rows (b, channel) x time,
//   coalesced row reads, the row staged once in LDS with its zero padding.
// Both brute-force kernels reuse the sliding-register-window structure of the shapelet kernels (ign_shapelet_*.h):
// lane <-> consecutive output times with the taps as wave-uniform SGPR operands (forward), lane <-> (filter, taps)
// with the row values as LDS broadcasts (gradient); fp32 FMA, deterministic two-stage reductions.
#include "ign_common.h"

typedef const __attribute__((address_space(4))) float* cfloat_p;

// ------------------------------------------------------------------------------------------------ sum (y1-mu)^2
// block = 128 threads = one (b,c) row at a time, TT = 8 outputs per lane, FT = 8 filters per pass.
constexpr int C1_TT = 8, C1_FT = 8, C1_J = 4, C1_THREADS = 128;

template <bool STORE_Y>
__device__ __forceinline__ void conv1_row(const float* xs, cfloat_p w, const float* mu, int T, int k1, int nf, int t0,
                                          float (&acc)[C1_FT][C1_TT]) {
    // acc[f][tt] = sum_j w[f*k1 + j] * xs[t0 + tt + j]   (xs already left-padded: xs[i] = xpad[i])
#pragma unroll
    for (int f = 0; f < C1_FT; ++f)
#pragma unroll
        for (int t = 0; t < C1_TT; ++t) acc[f][t] = 0.f;
    float xw[C1_TT + C1_J - 1];
#pragma unroll
    for (int i = 0; i < C1_TT - 1; ++i) xw[i] = xs[t0 + i];
    int j0 = 0;
    for (; j0 + C1_J <= k1; j0 += C1_J) {
#pragma unroll
        for (int jj = 0; jj < C1_J; ++jj) xw[C1_TT - 1 + jj] = xs[t0 + j0 + C1_TT - 1 + jj];
#pragma unroll
        for (int jj = 0; jj < C1_J; ++jj)
#pragma unroll
            for (int f = 0; f < C1_FT; ++f) {
                const float wv = (f < nf) ? w[f * k1 + j0 + jj] : 0.f;      // wave-uniform: SGPR operand
#pragma unroll
                for (int t = 0; t < C1_TT; ++t) acc[f][t] = fmaf(wv, xw[t + jj], acc[f][t]);
            }
#pragma unroll
        for (int i = 0; i < C1_TT - 1; ++i) xw[i] = xw[i + C1_J];
    }
    for (; j0 < k1; ++j0) {
        xw[C1_TT - 1] = xs[t0 + j0 + C1_TT - 1];
#pragma unroll
        for (int f = 0; f < C1_FT; ++f) {
            const float wv = (f < nf) ? w[f * k1 + j0] : 0.f;
#pragma unroll
            for (int t = 0; t < C1_TT; ++t) acc[f][t] = fmaf(wv, xw[t], acc[f][t]);
        }
#pragma unroll
        for (int i = 0; i < C1_TT - 1; ++i) xw[i] = xw[i + 1];
    }
}

__global__ void __launch_bounds__(C1_THREADS) conv1_sumsq_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                                     const float* __restrict__ mu, float* __restrict__ part,
                                                                     int rows, int T, int F1, int k1, int pl, int xs_len,
                                                                     int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float xs[];          // xpad of the current row
    __shared__ float red[C1_THREADS / 64][C1_FT];
    const int tid = threadIdx.x;
    const int f0 = blockIdx.y * C1_FT;
    const int nf = min(C1_FT, F1 - f0);
    const cfloat_p w = (cfloat_p)(uintptr_t)(w1 + (size_t)f0 * k1);
    float muf[C1_FT], s2[C1_FT];
#pragma unroll
    for (int f = 0; f < C1_FT; ++f) { muf[f] = (f < nf) ? mu[f0 + f] : 0.f; s2[f] = 0.f; }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int r = r0; r < r1; ++r) {
        __syncthreads();
        const float* row = x + (size_t)r * T;
        for (int i = tid; i < xs_len; i += C1_THREADS) {
            const int s = i - pl;
            xs[i] = (s >= 0 && s < T) ? row[s] : 0.f;
        }
        __syncthreads();
        for (int t0 = tid * C1_TT; t0 < T; t0 += C1_THREADS * C1_TT) {
            float acc[C1_FT][C1_TT];
            conv1_row<false>(xs, w, muf, T, k1, nf, t0, acc);
#pragma unroll
            for (int f = 0; f < C1_FT; ++f)
#pragma unroll
                for (int t = 0; t < C1_TT; ++t)
                    if (t0 + t < T) {
                        const float dv = acc[f][t] - muf[f];
                        s2[f] = fmaf(dv, dv, s2[f]);
                    }
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int f = 0; f < C1_FT; ++f) {
        float v = s2[f];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][f] = v;
    }
    __syncthreads();
    if (tid < nf) {
        float v = 0.f;
        for (int wv = 0; wv < C1_THREADS / 64; ++wv) v += red[wv][tid];
        part[(size_t)blockIdx.x * F1 + f0 + tid] = v;
    }
}

// ------------------------------------------------------------------------------------------------ gradient of the above
// G[f,j] = sum_{rows,t} (y1[f,t] - mu_f) * xpad[t + j].  Block = 256 threads; per row: phase (a) recompute the centred
// y1 row tile into LDS (thread <-> 4 consecutive t, all 8 filters), phase (b) lane <-> (f, 4 consecutive taps) slides
// over t with the y1 values as LDS broadcasts and a ping-pong register window on xpad.
constexpr int G_THREADS = 256, G_JJ = 4, G_TA = 4;

__global__ void __launch_bounds__(G_THREADS) conv1_sumsq_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                                    const float* __restrict__ mu, float* __restrict__ part,
                                                                    int rows, int T, int F1, int k1, int pl, int xs_len,
                                                                    int tpad, int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                       // [xs_len]   xpad (zero beyond the row)
    float* ys = smem + xs_len;              // [C1_FT][tpad]  centred y1 of the row (zero beyond T)
    const int tid = threadIdx.x;
    const int f0 = blockIdx.y * C1_FT;
    const int nf = min(C1_FT, F1 - f0);
    const cfloat_p w = (cfloat_p)(uintptr_t)(w1 + (size_t)f0 * k1);
    const int cpf = (k1 + G_JJ - 1) / G_JJ;                 // tap chunks per filter
    int fl = tid / cpf;
    const int jc = tid - fl * cpf;
    const bool active = fl < nf;
    if (!active) fl = nf - 1;
    const int jbase = jc * G_JJ;
    float acc[G_JJ];
#pragma unroll
    for (int q = 0; q < G_JJ; ++q) acc[q] = 0.f;

    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    for (int r = r0; r < r1; ++r) {
        __syncthreads();
        const float* row = x + (size_t)r * T;
        for (int i = tid; i < xs_len; i += G_THREADS) {
            const int s = i - pl;
            xs[i] = (s >= 0 && s < T) ? row[s] : 0.f;
        }
        __syncthreads();
        // (a) centred y1 tile
        for (int t0 = tid * G_TA; t0 < tpad; t0 += G_THREADS * G_TA) {
            float a[C1_FT][G_TA];
#pragma unroll
            for (int f = 0; f < C1_FT; ++f)
#pragma unroll
                for (int t = 0; t < G_TA; ++t) a[f][t] = 0.f;
            if (t0 < T) {
                float xw[G_TA + C1_J - 1];
#pragma unroll
                for (int i = 0; i < G_TA - 1; ++i) xw[i] = xs[t0 + i];
                int j0 = 0;
                for (; j0 + C1_J <= k1; j0 += C1_J) {
#pragma unroll
                    for (int jj = 0; jj < C1_J; ++jj) xw[G_TA - 1 + jj] = xs[t0 + j0 + G_TA - 1 + jj];
#pragma unroll
                    for (int jj = 0; jj < C1_J; ++jj)
#pragma unroll
                        for (int f = 0; f < C1_FT; ++f) {
                            const float wv = (f < nf) ? w[f * k1 + j0 + jj] : 0.f;
#pragma unroll
                            for (int t = 0; t < G_TA; ++t) a[f][t] = fmaf(wv, xw[t + jj], a[f][t]);
                        }
#pragma unroll
                    for (int i = 0; i < G_TA - 1; ++i) xw[i] = xw[i + C1_J];
                }
                for (; j0 < k1; ++j0) {
                    xw[G_TA - 1] = xs[t0 + j0 + G_TA - 1];
#pragma unroll
                    for (int f = 0; f < C1_FT; ++f) {
                        const float wv = (f < nf) ? w[f * k1 + j0] : 0.f;
#pragma unroll
                        for (int t = 0; t < G_TA; ++t) a[f][t] = fmaf(wv, xw[t], a[f][t]);
                    }
#pragma unroll
                    for (int i = 0; i < G_TA - 1; ++i) xw[i] = xw[i + 1];
                }
            }
#pragma unroll
            for (int f = 0; f < C1_FT; ++f) {
                float4 v;
                const float m = (f < nf) ? mu[f0 + f] : 0.f;
                v.x = (t0 + 0 < T) ? a[f][0] - m : 0.f;
                v.y = (t0 + 1 < T) ? a[f][1] - m : 0.f;
                v.z = (t0 + 2 < T) ? a[f][2] - m : 0.f;
                v.w = (t0 + 3 < T) ? a[f][3] - m : 0.f;
                *reinterpret_cast<float4*>(ys + f * tpad + t0) = v;
            }
        }
        __syncthreads();
        // (b) correlate: acc[q] += sum_t y[fl][t] * xpad[t + jbase + q]
        {
            const float* yl = ys + fl * tpad;
            const float* xl = xs + jbase;
            float Wa[G_JJ], Wb[G_JJ], A[G_JJ];
            float4 v = *reinterpret_cast<const float4*>(xl);
            Wa[0] = v.x; Wa[1] = v.y; Wa[2] = v.z; Wa[3] = v.w;
            for (int t = 0; t < tpad; t += 2 * G_JJ) {
                v = *reinterpret_cast<const float4*>(xl + t + G_JJ);
                Wb[0] = v.x; Wb[1] = v.y; Wb[2] = v.z; Wb[3] = v.w;
                v = *reinterpret_cast<const float4*>(yl + t);
                A[0] = v.x; A[1] = v.y; A[2] = v.z; A[3] = v.w;
#pragma unroll
                for (int tt = 0; tt < G_JJ; ++tt)
#pragma unroll
                    for (int q = 0; q < G_JJ; ++q)
                        acc[q] = fmaf(A[tt], (tt + q) < G_JJ ? Wa[(tt + q) % G_JJ] : Wb[(tt + q) % G_JJ], acc[q]);
                v = *reinterpret_cast<const float4*>(xl + t + 2 * G_JJ);
                Wa[0] = v.x; Wa[1] = v.y; Wa[2] = v.z; Wa[3] = v.w;
                v = *reinterpret_cast<const float4*>(yl + t + G_JJ);
                A[0] = v.x; A[1] = v.y; A[2] = v.z; A[3] = v.w;
#pragma unroll
                for (int tt = 0; tt < G_JJ; ++tt)
#pragma unroll
                    for (int q = 0; q < G_JJ; ++q)
                        acc[q] = fmaf(A[tt], (tt + q) < G_JJ ? Wb[(tt + q) % G_JJ] : Wa[(tt + q) % G_JJ], acc[q]);
            }
        }
    }
    if (active) {
        float* out = part + ((size_t)blockIdx.x * F1 + f0 + fl) * k1;
#pragma unroll
        for (int q = 0; q < G_JJ; ++q)
            if (jbase + q < k1) out[jbase + q] = acc[q];
    }
}

// ------------------------------------------------------------------------------------------------ depthwise 1-D conv
// y[r, t] = sum_j w[ch(r), j] * xpad[r, t + j],  rows r = (b, ch), ch = r % Cc;  'same' zero padding pl / k-1-pl.
// FLIP selects the correlation with the reversed filter (gradient w.r.t. the input: pl' = k-1-pl).
template <bool FLIP>
__global__ void __launch_bounds__(256) dwconv1d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, int rows, int Cc, int T, int k, int pl,
                                                       int xs_len) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                  // [xs_len]
    float* ws = sm + xs_len;         // [k]
    const int r = blockIdx.x;
    const int ch = r % Cc;
    const float* row = x + (size_t)r * T;
    for (int i = threadIdx.x; i < xs_len; i += 256) {
        const int s = i - pl;
        xs[i] = (s >= 0 && s < T) ? row[s] : 0.f;
    }
    for (int j = threadIdx.x; j < k; j += 256) ws[j] = FLIP ? w[(size_t)ch * k + (k - 1 - j)] : w[(size_t)ch * k + j];
    __syncthreads();
    for (int t0 = threadIdx.x * 4; t0 < T; t0 += 1024) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        float x0 = xs[t0], x1 = xs[t0 + 1], x2 = xs[t0 + 2];
        for (int j = 0; j < k; ++j) {
            const float x3 = xs[t0 + j + 3];
            const float wv = ws[j];
            a0 = fmaf(wv, x0, a0); a1 = fmaf(wv, x1, a1); a2 = fmaf(wv, x2, a2); a3 = fmaf(wv, x3, a3);
            x0 = x1; x1 = x2; x2 = x3;
        }
        float* yo = y + (size_t)r * T + t0;
        if (t0 + 3 < T) { yo[0] = a0; yo[1] = a1; yo[2] = a2; yo[3] = a3; }
        else { if (t0 < T) yo[0] = a0; if (t0 + 1 < T) yo[1] = a1; if (t0 + 2 < T) yo[2] = a2; }
    }
}

// dw[ch, j] partial over a slice of the batch: part[bs, ch, j] = sum_{b in slice} sum_t dy[b,ch,t] * xpad[b,ch,t + j]
__global__ void __launch_bounds__(256) dwconv1d_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ part, int B, int Cc, int T, int k, int pl,
                                                             int xs_len, int nbs) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                  // [xs_len]
    float* gs = sm + xs_len;         // [T]
    const int ch = blockIdx.x, bs = blockIdx.y;
    const int b0 = (int)(((long long)B * bs) / nbs), b1 = (int)(((long long)B * (bs + 1)) / nbs);
    const int j = threadIdx.x;       // taps beyond 256 loop below
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                  // taps j, j+256, j+512, j+768 (k <= 1024)
    for (int b = b0; b < b1; ++b) {
        __syncthreads();
        const float* row = x + ((size_t)b * Cc + ch) * T;
        const float* grow = dy + ((size_t)b * Cc + ch) * T;
        for (int i = threadIdx.x; i < xs_len; i += 256) {
            const int s = i - pl;
            xs[i] = (s >= 0 && s < T) ? row[s] : 0.f;
        }
        for (int i = threadIdx.x; i < T; i += 256) gs[i] = grow[i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jj = j + 256 * q;
            if (jj < k) {
                float a = acc[q];
                for (int t = 0; t < T; ++t) a = fmaf(gs[t], xs[t + jj], a);
                acc[q] = a;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int jj = j + 256 * q;
        if (jj < k) part[((size_t)bs * Cc + ch) * k + jj] = acc[q];
    }
}

// ------------------------------------------------------------------------------------------------ round 2: register-tiled forms
// The two kernels above read LDS once or twice per FMA (dwconv1d_kernel: x and w from LDS for 4 outputs; dwconv1d_bwd_w_kernel and
// autocorr_kernel: both operands from LDS for ONE FMA) and ran at 8-13 T FMA/s.  These keep the operands in registers:
//   dwconv1d_wave_kernel : one wave per (b, channel) row, lane <-> TT consecutive outputs with a sliding register window, the taps
//                          wave-uniform SGPR operands (the shapelet forward's main loop with an FMA): 1 LDS read per TT FMAs;
//   xcorr_kernel         : out[g][j] = sum_{rows of group g} sum_t a[r,t] * bpad[r, t + j]  -- the depthwise weight gradient (a = dy,
//                          b = x, group = channel) and the input's lag sums (a = b = x, one group).  Lane <-> 4 consecutive t, lags
//                          in tiles of 32: 128 FMAs per 10 ds_read_b128, all NT*32 accumulators live across the block's rows, ONE
//                          butterfly reduction per block.  Partials per block, added in ascending order afterwards.
template <int TT, bool FLIP>
__global__ void __launch_bounds__(64) dwconv1d_wave_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, int Cc, int T, int k, int pl, int xs_len) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    typedef const __attribute__((address_space(4))) float* cfp;
    // Lane l reads sample TT l + c of the padded row for the same c in every lane: stored plainly, the lanes of a half-wave would
    // hit 32 / TT banks (TT = 16: two banks, a 16-way conflict per ds_read_b32 -- the LDS reads then cost as much as the FMAs).
    // Sample e lives at e + e / TT instead: the lane pitch becomes TT + 1, odd, one bank per lane.
    constexpr int SH = TT == 16 ? 4 : TT == 8 ? 3 : 2;
#define IGN_SK(e) ((e) + ((e) >> SH))
    const int r = blockIdx.x, lane = threadIdx.x;
    const int ch = r % Cc;
    const float* row = x + (size_t)r * T;
    for (int i = lane; i < xs_len; i += 64) {
        const int s = i - pl;
        xs[IGN_SK(i)] = (s >= 0 && s < T) ? row[s] : 0.f;
    }
    __syncthreads();
    const cfp wk = (cfp)(uintptr_t)(w + (size_t)ch * k);
    const float* xl = xs + lane * (TT + 1);         // = IGN_SK(TT lane + c) - (c + c / TT)
    constexpr int J = 8;                            // taps per unrolled group: one s_load_dwordx8 of taps, TT-1 register moves
    float acc[TT], xw[TT + J - 1];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = 0.f;
#pragma unroll
    for (int i = 0; i < TT - 1; ++i) xw[i] = xl[IGN_SK(i)];
    int j0 = 0;
    for (; j0 + J <= k; j0 += J) {
#pragma unroll
        for (int jj = 0; jj < J; ++jj) xw[TT - 1 + jj] = xl[IGN_SK(j0 + TT - 1 + jj)];
#pragma unroll
        for (int jj = 0; jj < J; ++jj) {
            const float wv = FLIP ? wk[k - 1 - (j0 + jj)] : wk[j0 + jj];             // wave-uniform: s_load, SGPR operand
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[t] = fmaf(wv, xw[t + jj], acc[t]);
        }
#pragma unroll
        for (int i = 0; i < TT - 1; ++i) xw[i] = xw[i + J];
    }
    for (; j0 < k; ++j0) {
        xw[TT - 1] = xl[IGN_SK(j0 + TT - 1)];
        const float wv = FLIP ? wk[k - 1 - j0] : wk[j0];
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[t] = fmaf(wv, xw[t], acc[t]);
#pragma unroll
        for (int i = 0; i < TT - 1; ++i) xw[i] = xw[i + 1];
    }
    // lane-major registers -> time-major LDS -> coalesced stores (the row is no longer needed)
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TT; ++t) xs[lane * (TT + 1) + t] = acc[t];
    __syncthreads();
    float* yo = y + (size_t)r * T;
    for (int i = lane; i < T; i += 64) yo[i] = xs[IGN_SK(i)];
#undef IGN_SK
}

constexpr int XC_ROWS = 32;                        // rows per block and group
constexpr int XC_NA = 4, XC_NB = 6;                // register prefetch of one pass: 256*XC_NA floats of a rows, 256*XC_NB of b rows
template <int NT>
__global__ void __launch_bounds__(256) xcorr_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part,
                                                    int nrows_g /* rows per group */, int Cc, int T, int K, int pl, int lpr, int T4,
                                                    int BL, float* __restrict__ asum_part /* nullable: per-block sum of the a rows */) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int rpp = 256 / lpr;                      // rows per pass
    float* as_ = sm;                                // [rpp][T4]
    float* bs_ = sm + rpp * T4;                     // [rpp][BL]   bs_[i] = b[i - pl] (0 outside the row)
    const int ch = blockIdx.x, bsl = blockIdx.y;
    const int tid = threadIdx.x, slot = tid / lpr, li = tid - slot * lpr;
    const int u0 = 4 * li;
    const int r0 = bsl * XC_ROWS, r1 = min(nrows_g, r0 + XC_ROWS);
    float acc[NT][32];
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[q][j] = 0.f;
    // rows are prefetched into registers one pass ahead: the global-load latency of pass p+1 runs under the FMAs of pass p
    constexpr int NA = XC_NA, NB = XC_NB;           // one pass = rpp*T4 <= 256 NA and rpp*BL <= 256 NB floats: xcorr_geometry picks lpr so
    float pa[NA], pb[NB];
    auto gload = [&](int rb) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int i = tid + 256 * q;
            const int sl = i / T4, u = i - sl * T4;
            const int rr = rb + sl;
            pa[q] = (i < rpp * T4 && rr < r1 && u < T) ? a[((size_t)rr * Cc + ch) * T + u] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int i = tid + 256 * q;
            const int sl = i / BL, u = i - sl * BL - pl;
            const int rr = rb + sl;
            pb[q] = (i < rpp * BL && rr < r1 && u >= 0 && u < T) ? b[((size_t)rr * Cc + ch) * T + u] : 0.f;
        }
    };
    if (r0 < r1) gload(r0);
    float asum = 0.f;                               // every a element of the block passes through pa exactly once (0 where masked)
    for (int rb = r0; rb < r1; rb += rpp) {
        __syncthreads();                            // the previous pass has consumed the LDS rows
#pragma unroll
        for (int q = 0; q < NA; ++q) { const int i = tid + 256 * q; if (i < rpp * T4) as_[i] = pa[q]; asum += pa[q]; }
#pragma unroll
        for (int q = 0; q < NB; ++q) { const int i = tid + 256 * q; if (i < rpp * BL) bs_[i] = pb[q]; }
        __syncthreads();
        if (rb + rpp < r1) gload(rb + rpp);
        if (u0 < T4) {
            const float4 av = *reinterpret_cast<const float4*>(as_ + slot * T4 + u0);
            const float aq[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                float wv[36];
                const float* wp = bs_ + slot * BL + u0 + 32 * q;
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    const float4 t4 = *reinterpret_cast<const float4*>(wp + 4 * i);
                    wv[4 * i] = t4.x; wv[4 * i + 1] = t4.y; wv[4 * i + 2] = t4.z; wv[4 * i + 3] = t4.w;
                }
#pragma unroll
                for (int j = 0; j < 32; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[q][j] = fmaf(aq[e], wv[e + j], acc[q][j]);
            }
        }
    }
    // one reduction per block: butterfly inside each wave (fixed pattern), then the four waves in ascending order
    __syncthreads();
    float* red = sm;                                // [4][NT*32]
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            float v = acc[q][j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) red[wave * NT * 32 + q * 32 + j] = v;
        }
    __syncthreads();
    if (tid < NT * 32 && tid < K)
        part[((size_t)bsl * Cc + ch) * K + tid] = ((red[tid] + red[NT * 32 + tid]) + red[2 * NT * 32 + tid]) + red[3 * NT * 32 + tid];
    if (asum_part) {                                // block-uniform
        __syncthreads();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) asum += __shfl_xor(asum, o, 64);
        if (lane == 0) red[wave] = asum;
        __syncthreads();
        if (tid == 0) asum_part[(size_t)bsl * Cc + ch] = ((red[0] + red[1]) + red[2]) + red[3];
    }
}

// Edge terms of the window Gram matrix (models/eegcnn.py::_window_gram): per row only the first / last k-1 samples of the
// zero-padded row xp (left pad pl) are involved,
//     out[0][s][d] = sum_rows xp[s] xp[s+d]          (s, s+d < m = k-1: the head),
//     out[1][s][d] = sum_rows xp[T+s] xp[T+s+d]      (the tail),
// which round 1 took from two batched GEMMs over zero-padded copies.  Thread <-> (lag d, half of the s range): 2 x 62
// accumulators, the two segments of a row in LDS; partials per block, the caller adds them in double.  4.8e8 FMA at the benchmark
// shape; eight rows are staged per barrier pair.
constexpr int EG_MAXM = 124, EG_HALF = 62;
constexpr int EG_RB = 8;                            // rows staged per barrier pair: 16 independent loads per thread in flight (one row per
                                                    // pair made the kernel a chain of ~120 exposed memory round trips: 0.18 ms)
__global__ void __launch_bounds__(256) edge_lagprod_kernel(const float* __restrict__ x, float* __restrict__ part, int rows, int T, int k,
                                                           int pl, int rows_per_block) {
    constexpr int W = 2 * EG_MAXM + 8;              // = 256 staged samples per row and end: one per thread
    __shared__ float hs[EG_RB][W], ts[EG_RB][W];
    const int m = k - 1;
    const int d = threadIdx.x & 127, half = threadIdx.x >> 7;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float ah[EG_HALF], at[EG_HALF];
#pragma unroll
    for (int i = 0; i < EG_HALF; ++i) { ah[i] = 0.f; at[i] = 0.f; }
    float csh = 0.f, cst = 0.f;            // thread i < m: column sums of xp[i] and xp[T + i] over the block's rows (column 127)
    const int i = threadIdx.x;
    // head: xp[i] = x[i - pl], i < m;   tail: xp[T + i] = x[T + i - pl], i < pl   (zero elsewhere, and beyond m)
    const int sh = i - pl, st = T + i - pl;
    const bool hok = i < m && sh >= 0 && sh < T, tok = i < m && st >= 0 && st < T;
    for (int rb = r0; rb < r1; rb += EG_RB) {
        float vh[EG_RB], vt[EG_RB];
#pragma unroll
        for (int q = 0; q < EG_RB; ++q) {
            const int r = rb + q;
            vh[q] = (hok && r < r1) ? x[(size_t)r * T + sh] : 0.f;
            vt[q] = (tok && r < r1) ? x[(size_t)r * T + st] : 0.f;
        }
        __syncthreads();                            // the previous batch of rows has been consumed
#pragma unroll
        for (int q = 0; q < EG_RB; ++q) { hs[q][i] = vh[q]; ts[q][i] = vt[q]; }
        __syncthreads();
        const int nq = min(EG_RB, r1 - rb);
        for (int q = 0; q < nq; ++q) {              // rows in ascending order: the sums are those of the one-row-per-pass kernel, bit for bit
#pragma unroll
            for (int e = 0; e < EG_HALF; ++e) {
                const int s = half * EG_HALF + e;
                ah[e] = fmaf(hs[q][s], hs[q][s + d], ah[e]);
                at[e] = fmaf(ts[q][s], ts[q][s + d], at[e]);
            }
            if (threadIdx.x < EG_MAXM) { csh += hs[q][threadIdx.x]; cst += ts[q][threadIdx.x]; }
        }
    }
    if (threadIdx.x < EG_MAXM) {
        float* pb = part + (size_t)blockIdx.x * 2 * EG_MAXM * 128;
        pb[(size_t)threadIdx.x * 128 + 127] = csh;
        pb[(size_t)(EG_MAXM + threadIdx.x) * 128 + 127] = cst;
    }
    if (d < k) {
        float* pb = part + (size_t)blockIdx.x * 2 * EG_MAXM * 128;
#pragma unroll
        for (int e = 0; e < EG_HALF; ++e) {
            const int s = half * EG_HALF + e;
            pb[(size_t)s * 128 + d] = ah[e];
            pb[(size_t)(EG_MAXM + s) * 128 + d] = at[e];
        }
    }
}

// launch geometry of xcorr_kernel for rows of T samples and K lags; returns 0 when the shape is outside its tile
static int xcorr_geometry(int T, int K, int* lpr, int* T4, int* BL, int* NT, size_t* lds) {
    if (T > 1024 || K > 128 || T < 1 || K < 1) return 0;
    *T4 = (T + 3) & ~3;
    *NT = K <= 32 ? 1 : 4;
    // lanes per row: the smallest power of two that covers the row AND keeps one pass of 256 / lpr rows inside the kernel's
    // register prefetch (rpp*T4 <= 256 XC_NA, rpp*BL <= 256 XC_NB).  Short rows would otherwise put more rows into a pass than
    // the prefetch loops cover (T <= 64 at K <= 32, T <= 128 or 233..256 at K > 32) and leave the tail slots of bs_ unwritten.
    int l = 1;
    for (;; l <<= 1) {
        const int bl = 4 * l + 32 * (*NT) + 8;      // the last lane's last window: u0 + 32 (NT-1) + 36 <= 4 lpr + 32 NT + 4
        const int rpp_ = 256 / l;
        if (l >= *T4 / 4 && rpp_ * (*T4) <= 256 * XC_NA && rpp_ * bl <= 256 * XC_NB) break;
        if (l == 256) return 0;
    }
    *lpr = l;
    *BL = 4 * l + 32 * (*NT) + 8;
    const int rpp = 256 / l;
    *lds = std::max((size_t)rpp * (*T4 + *BL), (size_t)4 * (*NT) * 32) * sizeof(float);
    return 1;
}

// ------------------------------------------------------------------------------------------------ C ABI
static int conv1_check(const char* who, const void* x, const void* w1, const void* mu, const void* out, const void* ws,
                       int rows, int T, int F1, int k1, int pl) {
    if (!x || !w1 || !mu || !out || !ws || rows <= 0 || T <= 0 || F1 <= 0 || k1 <= 0 || pl < 0 || pl >= k1) {
        ign_set_error("%s: null pointer or bad dimension (rows=%d T=%d F1=%d k1=%d pl=%d)", who, rows, T, F1, k1, pl);
        return IGN_E_ARG;
    }
    if ((size_t)(T + k1 + 64) * 4 * 10 > 160 * 1024) {
        ign_set_error("%s: T=%d does not fit the LDS row tile", who, T);
        return IGN_E_TOOBIG;
    }
    return 0;
}

static int conv1_blocks(int rows, int* rows_per_block) {
    int rpb = (rows + 2047) / 2048;                  // ~2048 blocks: fills 256 CUs with a short tail
    if (rpb < 1) rpb = 1;
    *rows_per_block = rpb;
    return (rows + rpb - 1) / rpb;
}

extern "C" size_t ign_conv1_sumsq_workspace_bytes(int rows, int F1, int k1) {
    if (rows <= 0 || F1 <= 0 || k1 <= 0) return 0;
    int rpb;
    const int nb = conv1_blocks(rows, &rpb);
    return (size_t)nb * F1 * k1 * sizeof(float);     // sized for the gradient; the forward uses nb*F1 of it
}

extern "C" int ign_conv1_sumsq_fwd(const float* x, const float* w1, const float* mu, float* m2, void* workspace, int rows,
                                   int T, int F1, int k1, int pad_left, void* stream) {
    static const char* who = "ign_conv1_sumsq_fwd";
    int rc;
    if ((rc = conv1_check(who, x, w1, mu, m2, workspace, rows, T, F1, k1, pad_left))) return rc;
    int rpb;
    const int nb = conv1_blocks(rows, &rpb);
    const int tslots = ((T + C1_TT - 1) / C1_TT) * C1_TT;
    const int xs_len = (tslots + k1 + C1_TT + 3) & ~3;
    hipStream_t s = (hipStream_t)stream;
    {
        IgnScopedTimer tm("conv1_sumsq_fwd", s);
        hipLaunchKernelGGL(conv1_sumsq_fwd_kernel, dim3(nb, (F1 + C1_FT - 1) / C1_FT), dim3(C1_THREADS), (size_t)xs_len * 4, s, x,
                           w1, mu, (float*)workspace, rows, T, F1, k1, pad_left, xs_len, rpb);
    }
    if ((rc = ign_check_launch("conv1_sumsq_fwd_kernel"))) return rc;
    ign_launch_reduce_parts((const float*)workspace, m2, nb, (size_t)F1, s);
    return ign_check_launch("reduce_parts_kernel");
}

extern "C" int ign_conv1_sumsq_bwd(const float* x, const float* w1, const float* mu, float* g_fj, void* workspace, int rows,
                                   int T, int F1, int k1, int pad_left, void* stream) {
    static const char* who = "ign_conv1_sumsq_bwd";
    int rc;
    if ((rc = conv1_check(who, x, w1, mu, g_fj, workspace, rows, T, F1, k1, pad_left))) return rc;
    if (((k1 + G_JJ - 1) / G_JJ) * std::min(F1, C1_FT) > G_THREADS) {
        ign_set_error("%s: F1*k1 = %d*%d exceeds the %d (filter, tap-chunk) lanes of a block", who, F1, k1, G_THREADS);
        return IGN_E_UNSUP;
    }
    int rpb;
    const int nb = conv1_blocks(rows, &rpb);
    const int tpad = ((T + 2 * G_JJ - 1) / (2 * G_JJ)) * (2 * G_JJ);
    const int xs_len = (tpad + k1 + 2 * G_JJ + 8 + 3) & ~3;
    const size_t lds = ((size_t)xs_len + (size_t)C1_FT * tpad) * 4;
    hipStream_t s = (hipStream_t)stream;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)conv1_sumsq_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    {
        IgnScopedTimer tm("conv1_sumsq_bwd", s);
        hipLaunchKernelGGL(conv1_sumsq_bwd_kernel, dim3(nb, (F1 + C1_FT - 1) / C1_FT), dim3(G_THREADS), lds, s, x, w1, mu,
                           (float*)workspace, rows, T, F1, k1, pad_left, xs_len, tpad, rpb);
    }
    if ((rc = ign_check_launch("conv1_sumsq_bwd_kernel"))) return rc;
    ign_launch_reduce_parts((const float*)workspace, g_fj, nb, (size_t)F1 * k1, s);
    return ign_check_launch("reduce_parts_kernel");
}

static int dw_check(const char* who, const void* a, const void* b, const void* c, int B, int Cc, int T, int k, int pl) {
    if (!a || !b || !c || B <= 0 || Cc <= 0 || T <= 0 || k <= 0 || k > 1024 || pl < 0 || pl >= k) {
        ign_set_error("%s: null pointer or bad dimension (B=%d C=%d T=%d k=%d pl=%d)", who, B, Cc, T, k, pl);
        return IGN_E_ARG;
    }
    if ((size_t)(2 * T + 2 * k + 16) * 4 > 64 * 1024) {
        ign_set_error("%s: T=%d k=%d does not fit the LDS row tile", who, T, k);
        return IGN_E_TOOBIG;
    }
    return 0;
}

extern "C" int ign_dwconv1d_fwd(const float* x, const float* w, float* y, int B, int Cc, int T, int k, int pad_left,
                                int flip, void* stream) {
    int rc;
    if ((rc = dw_check("ign_dwconv1d_fwd", x, w, y, B, Cc, T, k, pad_left))) return rc;
    hipStream_t s = (hipStream_t)stream;
    IgnScopedTimer tm("dwconv1d", s);
    if (T <= 1024) {
        // one wave per row, TT = 4 / 8 / 16 outputs per lane (the smallest that covers T)
        const int TT = T <= 256 ? 4 : T <= 512 ? 8 : 16;
        const int xl = (64 * TT + k + 3) & ~3;
        const size_t l2 = (size_t)(xl + xl / TT + 4) * 4;             // skewed layout: sample e at e + e / TT

        const dim3 grid((unsigned)B * Cc), block(64);
#define IGN_DW(TTV)                                                                                                                 \
        do {                                                                                                                        \
            if (flip) hipLaunchKernelGGL((dwconv1d_wave_kernel<TTV, true>), grid, block, l2, s, x, w, y, Cc, T, k, pad_left, xl);   \
            else      hipLaunchKernelGGL((dwconv1d_wave_kernel<TTV, false>), grid, block, l2, s, x, w, y, Cc, T, k, pad_left, xl);  \
        } while (0)
        if (TT == 4) IGN_DW(4); else if (TT == 8) IGN_DW(8); else IGN_DW(16);
#undef IGN_DW
        return ign_check_launch("dwconv1d_wave_kernel");
    }
    const int xs_len = (T + k + 8 + 3) & ~3;
    const size_t lds = ((size_t)xs_len + k) * 4;
    if (flip) hipLaunchKernelGGL(dwconv1d_kernel<true>, dim3((unsigned)B * Cc), dim3(256), lds, s, x, w, y, B * Cc, Cc, T, k, pad_left, xs_len);
    else      hipLaunchKernelGGL(dwconv1d_kernel<false>, dim3((unsigned)B * Cc), dim3(256), lds, s, x, w, y, B * Cc, Cc, T, k, pad_left, xs_len);
    return ign_check_launch("dwconv1d_kernel");
}

extern "C" size_t ign_dwconv1d_bwd_weight_workspace_bytes(int B, int Cc, int k) {
    if (B <= 0 || Cc <= 0 || k <= 0) return 0;
    // slabs of per-block partials: min(B, 32) batch slices (dwconv1d_bwd_w_kernel) or ceil(B / XC_ROWS) blocks of rows (xcorr_kernel)
    const int nbs = std::max(std::max(1, std::min(B, 32)), (B + XC_ROWS - 1) / XC_ROWS);
    return (size_t)nbs * Cc * k * sizeof(float);
}

extern "C" int ign_dwconv1d_bwd_weight(const float* x, const float* dy, float* dw, void* workspace, int B, int Cc, int T,
                                       int k, int pad_left, void* stream) {
    int rc;
    if ((rc = dw_check("ign_dwconv1d_bwd_weight", x, dy, dw, B, Cc, T, k, pad_left))) return rc;
    if (!workspace) { ign_set_error("ign_dwconv1d_bwd_weight: null workspace"); return IGN_E_ARG; }
    hipStream_t s = (hipStream_t)stream;
    {
        int lpr, T4, BL, NT; size_t l2;
        if (xcorr_geometry(T, k, &lpr, &T4, &BL, &NT, &l2)) {
            // dw[ch][j] = sum_{b,t} dy[b,ch,t] xpad[b,ch,t+j]: the cross-correlation kernel, group = channel, 32 samples per block
            const int nsl = (B + XC_ROWS - 1) / XC_ROWS;              // slabs: ign_dwconv1d_bwd_weight_workspace_bytes reserves max(min(B,32), nsl)
            {
                IgnScopedTimer tm("dwconv1d_bwd_w", s);
                if (NT == 1) hipLaunchKernelGGL(xcorr_kernel<1>, dim3(Cc, nsl), dim3(256), l2, s, dy, x, (float*)workspace, B, Cc, T, k,
                                                pad_left, lpr, T4, BL, (float*)nullptr);
                else         hipLaunchKernelGGL(xcorr_kernel<4>, dim3(Cc, nsl), dim3(256), l2, s, dy, x, (float*)workspace, B, Cc, T, k,
                                                pad_left, lpr, T4, BL, (float*)nullptr);
            }
            if ((rc = ign_check_launch("xcorr_kernel"))) return rc;
            ign_launch_reduce_parts((const float*)workspace, dw, nsl, (size_t)Cc * k, s);
            return ign_check_launch("reduce_parts_kernel");
        }
    }
    const int nbs = std::max(1, std::min(B, 32));
    const int xs_len = (T + k + 8 + 3) & ~3;
    const size_t lds = ((size_t)xs_len + T) * 4;
    {
        IgnScopedTimer tm("dwconv1d_bwd_w", s);
        hipLaunchKernelGGL(dwconv1d_bwd_w_kernel, dim3(Cc, nbs), dim3(256), lds, s, x, dy, (float*)workspace, B, Cc, T, k, pad_left,
                           xs_len, nbs);
    }
    if ((rc = ign_check_launch("dwconv1d_bwd_w_kernel"))) return rc;
    ign_launch_reduce_parts((const float*)workspace, dw, nbs, (size_t)Cc * k, s);
    return ign_check_launch("reduce_parts_kernel");
}

// ------------------------------------------------------------------------------------------------ lag sums (autocorrelation)
// C[d] = sum_rows sum_u x[row][u] * x[row][u + d], d = 0 .. K-1 (K <= 128).
// BatchNorm-1 of the EEG-CNN block needs the batch variance of y[f] = w1[f] (*) x over (batch, electrode, time) -- the one
// statistic that touches the un-contracted (B, F1, C, T) convolution (IGN/model/eegcnn.py:90-91).  sum_t y[f,t]^2 is the
// quadratic form w1[f]^T G w1[f] with G[j,j'] = sum_{rows,t} xpad[t+j] xpad[t+j'], and G is this lag vector minus edge terms
// that only involve the first / last k-1 samples of each row (models/eegcnn.py: _window_gram).  The lag sums cost K*T FMA per
// row for ALL filters -- F1 times fewer than evaluating the convolutions (conv1_sumsq_fwd_kernel) -- and the gradient with
// respect to w1 becomes 2 G w1: the 6.2e10-FMA backward pass (conv1_sumsq_bwd_kernel) disappears from the step.
constexpr int AC_LAGS = 128;
constexpr int AC_ROWS = 16;                       // rows per block (two at a time: threads 0..127 / 128..255)

__global__ void __launch_bounds__(256) autocorr_kernel(const float* __restrict__ x, float* __restrict__ part, int rows, int T,
                                                       int K, int xs_len) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int slot = threadIdx.x >> 7, d = threadIdx.x & 127;
    float* xs = smem + slot * xs_len;
    const int T4 = (T + 3) & ~3;
    float acc = 0.f;
    for (int i = 0; i < AC_ROWS; i += 2) {
        const int r = blockIdx.x * AC_ROWS + i + slot;
        __syncthreads();
        for (int u = d; u < xs_len; u += AC_LAGS) xs[u] = (r < rows && u < T) ? x[(size_t)r * T + u] : 0.f;
        __syncthreads();
        if (d < K && r < rows) {
            const float* xd = xs + d;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int u = 0; u < T4; u += 4) {
                const float4 a = *reinterpret_cast<const float4*>(xs + u);            // same address in every lane: broadcast
                a0 = fmaf(a.x, xd[u], a0);
                a1 = fmaf(a.y, xd[u + 1], a1);
                a2 = fmaf(a.z, xd[u + 2], a2);
                a3 = fmaf(a.w, xd[u + 3], a3);
            }
            acc += (a0 + a1) + (a2 + a3);
        }
    }
    if (d < K) part[((size_t)blockIdx.x * 2 + slot) * K + d] = acc;
}

// partial rows the caller must provide (and sum): blocks of XC_ROWS rows for rows of <= 1024 samples (xcorr_kernel), else the
// 16-row blocks of autocorr_kernel; the count depends on T only through that switch, so the larger of the two is reserved and
// ign_autocorr_fwd zero-fills what it does not write
extern "C" long long ign_autocorr_parts(int rows) { return rows > 0 ? 2LL * ((rows + AC_ROWS - 1) / AC_ROWS) : 0; }

// part: (ign_autocorr_parts(rows), K) partial lag sums; the caller adds them up (in double)
extern "C" int ign_autocorr_fwd(const float* x, float* part, int rows, int T, int K, void* stream) {
    static const char* who = "ign_autocorr_fwd";
    if (!x || !part || rows <= 0 || T <= 0 || K <= 0) {
        ign_set_error("%s: bad argument (rows=%d T=%d K=%d)", who, rows, T, K);
        return IGN_E_ARG;
    }
    if (K > AC_LAGS) { ign_set_error("%s: K=%d lags exceed %d", who, K, AC_LAGS); return IGN_E_UNSUP; }
    {
        int lpr, T4, BL, NT; size_t l2;
        if (xcorr_geometry(T, K, &lpr, &T4, &BL, &NT, &l2)) {
            // C[d] = sum_rows sum_u x[u] x[u+d]: the cross-correlation of every row with itself, one group
            const int nsl = (rows + XC_ROWS - 1) / XC_ROWS;
            const long long nparts = ign_autocorr_parts(rows);
            hipStream_t s = (hipStream_t)stream;
            if (nparts > nsl) (void)hipMemsetAsync(part + (size_t)nsl * K, 0, (size_t)(nparts - nsl) * K * sizeof(float), s);
            IgnScopedTimer tm("autocorr", s);
            if (NT == 1) hipLaunchKernelGGL(xcorr_kernel<1>, dim3(1, nsl), dim3(256), l2, s, x, x, part, rows, 1, T, K, 0, lpr, T4, BL, (float*)nullptr);
            else         hipLaunchKernelGGL(xcorr_kernel<4>, dim3(1, nsl), dim3(256), l2, s, x, x, part, rows, 1, T, K, 0, lpr, T4, BL, (float*)nullptr);
            return ign_check_launch("xcorr_kernel");
        }
    }
    const int xs_len = (((T + 3) & ~3) + AC_LAGS + 4 + 3) & ~3;
    const size_t lds = (size_t)2 * xs_len * sizeof(float);
    if (lds > 64 * 1024) { ign_set_error("%s: T=%d needs %zu bytes of LDS", who, T, lds); return IGN_E_TOOBIG; }
    const int nb = (rows + AC_ROWS - 1) / AC_ROWS;
    IgnScopedTimer tm("autocorr", (hipStream_t)stream);
    hipLaunchKernelGGL(autocorr_kernel, dim3(nb), dim3(256), lds, (hipStream_t)stream, x, part, rows, T, K, xs_len);
    return ign_check_launch("autocorr_kernel");
}

// ign_autocorr_fwd on the register-tiled kernel only (rows of <= 1024 samples), which then also returns the plain sum of its rows per
// block: rowsum_part[p], p < *used_parts = the partial rows actually written (part rows beyond are NOT touched).
extern "C" int ign_autocorr_sum_fwd(const float* x, float* part, float* rowsum_part, int rows, int T, int K, int* used_parts,
                                    void* stream) {
    static const char* who = "ign_autocorr_sum_fwd";
    if (!x || !part || !rowsum_part || !used_parts || rows <= 0 || T <= 0 || K <= 0) {
        ign_set_error("%s: bad argument (rows=%d T=%d K=%d)", who, rows, T, K);
        return IGN_E_ARG;
    }
    int lpr, T4, BL, NT; size_t l2;
    if (K > AC_LAGS || !xcorr_geometry(T, K, &lpr, &T4, &BL, &NT, &l2)) {
        ign_set_error("%s: T=%d K=%d outside the register-tiled kernel (T <= 1024, K <= 128)", who, T, K);
        return IGN_E_UNSUP;
    }
    const int nsl = (rows + XC_ROWS - 1) / XC_ROWS;
    hipStream_t s = (hipStream_t)stream;
    IgnScopedTimer tm("autocorr", s);
    if (NT == 1) hipLaunchKernelGGL(xcorr_kernel<1>, dim3(1, nsl), dim3(256), l2, s, x, x, part, rows, 1, T, K, 0, lpr, T4, BL, rowsum_part);
    else         hipLaunchKernelGGL(xcorr_kernel<4>, dim3(1, nsl), dim3(256), l2, s, x, x, part, rows, 1, T, K, 0, lpr, T4, BL, rowsum_part);
    *used_parts = nsl;
    return ign_check_launch("xcorr_kernel");
}

// part: (ign_edge_lagprod_parts(rows), 2, 124, 128) floats; [.,0,s,d] head products, [.,1,s,d] tail products; column 127 (never a
// lag: k <= 125) carries the column sums [.,0,s,127] = sum_rows xp[s], [.,1,s,127] = sum_rows xp[T+s] (s >= k-1 rows and
// d >= k columns are zero / unwritten garbage-free: the kernel writes every (s < 124, d < k) slot).  k <= 125.
extern "C" long long ign_edge_lagprod_parts(int rows) { return rows > 0 ? std::min(256, rows) : 0; }

extern "C" int ign_edge_lagprod_fwd(const float* x, float* part, int rows, int T, int k, int pad_left, void* stream) {
    static const char* who = "ign_edge_lagprod_fwd";
    if (!x || !part || rows <= 0 || T <= 0 || k <= 1 || pad_left < 0 || pad_left >= k) {
        ign_set_error("%s: bad argument (rows=%d T=%d k=%d pl=%d)", who, rows, T, k, pad_left);
        return IGN_E_ARG;
    }
    if (k - 1 > EG_MAXM) { ign_set_error("%s: k=%d exceeds %d taps", who, k, EG_MAXM + 1); return IGN_E_UNSUP; }
    const int nb = (int)ign_edge_lagprod_parts(rows);
    const int rpb = (rows + nb - 1) / nb;
    IgnScopedTimer tm("edge_lagprod", (hipStream_t)stream);
    hipLaunchKernelGGL(edge_lagprod_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, part, rows, T, k, pad_left, rpb);
    return ign_check_launch("edge_lagprod_kernel");
}

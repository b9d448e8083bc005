// Shared declarations of the channels-last convolution GEMMs (ign_clconv_f32.hip, ign_clconv_x6.hip) and of the
// BatchNorm glue (ign_bn.hip): tile constants, the row map that turns a GEMM into a convolution, vector load / store
// helpers, the argument block of the NT kernels and their common epilogue.
#pragma once
#include "ign_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int acc_row16(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

constexpr int TM = 128, TN = 128, KC = 16;
constexpr int NT_PITCH = KC + 4;             // floats per staged row (k contiguous); 20*i mod 64 is conflict-free for b128
constexpr int TN_PITCH = 128 + 8;            // floats per staged reduction row (outputs contiguous)

// logical GEMM row m -> float offset of its first element in a (samples, rows, row_pitch) buffer
struct RowMap {
    int rows_logical;        // rows per sample in the GEMM's row index space
    int row0;                // first physical row of a sample that logical row 0 maps to (skips padding)
    int row_pitch;           // floats between consecutive rows
    long long sample_pitch;  // floats between samples
};
__device__ __forceinline__ long long row_off(const RowMap& rm, int m) {
    const int s = m / rm.rows_logical;
    const int r = m - s * rm.rows_logical;
    return (long long)s * rm.sample_pitch + (long long)(rm.row0 + r) * rm.row_pitch;
}

template <int V> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<4> { typedef float4 T; };

template <int V>
__device__ __forceinline__ void vload(float (&d)[V], const float* p) {
    if (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
    else if (V == 2) { const float2 t = *reinterpret_cast<const float2*>(p); d[0] = t.x; d[1] = t.y; }
    else d[0] = *p;
}
template <int V>
__device__ __forceinline__ void vstore(float* p, const float (&d)[V]) {
    if (V == 4) *reinterpret_cast<float4*>(p) = make_float4(d[0], d[1], d[2], d[3]);
    else if (V == 2) *reinterpret_cast<float2*>(p) = make_float2(d[0], d[1]);
    else *p = d[0];
}

enum { EPI_BIAS_STATS = 0, EPI_MASK_STATS = 1, EPI_GELU_BWD = 2 };

// d gelu(u) / du of the exact (erf) GELU, as aten::gelu_backward evaluates it: Phi(u) + u phi(u)
__device__ __forceinline__ float ign_gelu_grad(float u) {
    const float cdf = 0.5f * (1.f + erff(u * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * u * u);
    return fmaf(u, pdf, cdf);
}

struct GemmNTArgs {
    const float* A; RowMap am; int K;          // A[m][kk] = A[row_off(m) + kk]
    const float* Bt; int ldb;                  // Bt[n][kk]
    float* C; int M, N;                        // dense (M, N)
    const float* bias;                         // [N] or null
    const float* pro_a; const float* pro_b; int pro_c;    // prologue: A <- relu(pro_a[c]*A + pro_b[c]), c = kk % pro_c
    float* part;                               // (mtiles, 2, N) partial sums, or null
    // EPI_MASK_STATS: g = acc * [ea*y + eb > 0]; partials of g and g*(y - mean)*invstd.   EPI_GELU_BWD: C = acc * gelu'(ey) (ey = the
    // pre-activation u, same (M, N) layout as C): the backward of y = gelu(u) in the epilogue of the GEMM that produces dL/dy
    const float* ey; const float* ea; const float* eb; const float* emean; const float* einv;
    int mtiles, ntiles;
    const unsigned short* B3; int Kp;          // split-bf16 kernels: Bt as 3 bf16 planes (3, N, Kp), Kp = K rounded up to 8
    float* amax_out;                           // nullable: max |C| is max'ed into this device float (integer atomicMax on the bits)
};

// ---- epilogue shared by the fp32 and the split-bf16 kernels.  Lane (l31, h) of accumulator (i, j) holds column
// n = n0 + wn*64 + j*32 + l31 and the 16 rows m0 + wm*64 + i*32 + acc_row16(r, h).  `red` is >= 512 floats of LDS that no
// wave still reads.  FULL = the whole 128x128 tile is inside the output: no per-element masks, and the 16 y values an
// accumulator needs (EPI_MASK_STATS) are fetched as 16 independent loads before any of them is used.
// SC: the accumulators hold the product of two operands that were multiplied by powers of two before they were split (the
// two-plane fp16 path, see ign_pow2_scale): every value is multiplied by `osc` = 1 / (scale_a * scale_b) first (exact).
template <int EPI, bool FULL, bool SC = false>
__device__ __forceinline__ void nt_epilogue_body(const GemmNTArgs& a, const f32x16 (&acc)[2][2], float (&s1)[2], float (&s2)[2],
                                                 int m0, int n0, int m_lim, float osc = 1.f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    float amax = 0.f;                          // SC kernels only: magnitude of the output, for the GEMM that consumes it
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        const bool n_ok = FULL || n < a.N;
        const int nc = n_ok ? n : a.N - 1;
        const float bv = (EPI == EPI_BIAS_STATS && a.bias) ? a.bias[nc] : 0.f;
        float ea = 0.f, eb = 0.f, em = 0.f, ei = 0.f;
        if (EPI == EPI_MASK_STATS) { ea = a.ea[nc]; eb = a.eb[nc]; em = a.emean[nc]; ei = a.einv[nc]; }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mb = m0 + wm * 64 + i * 32;
            float yv[16];
            if (EPI == EPI_MASK_STATS || EPI == EPI_GELU_BWD) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + acc_row16(r, h);
                    const int mc = FULL ? m : min(m, m_lim - 1);
                    yv[r] = a.ey[(long long)mc * a.N + nc];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + acc_row16(r, h);
                const bool ok = FULL || (n_ok && m < m_lim);
                float v = SC ? acc[i][j][r] * osc : acc[i][j][r];
                if (EPI == EPI_BIAS_STATS) {
                    v += bv;
                    if (ok) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
                } else if (EPI == EPI_GELU_BWD) {
                    v *= ign_gelu_grad(yv[r]);
                } else {
                    v = (fmaf(ea, yv[r], eb) > 0.f) ? v : 0.f;
                    if (ok) { s1[j] += v; s2[j] = fmaf(v, (yv[r] - em) * ei, s2[j]); }
                }
                if (ok) a.C[(long long)m * a.N + n] = v;
                if (SC && ok) amax = fmaxf(amax, fabsf(v));
            }
        }
    }
    if (SC && a.amax_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
        if (lane == 0) ign_atomic_absmax(a.amax_out, amax);
    }
}

template <int EPI, bool SC = false>
__device__ __forceinline__ void nt_epilogue(const GemmNTArgs& a, const f32x16 (&acc)[2][2], float* red, int mt, int m0, int n0,
                                            int m_lim, float osc = 1.f) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    if (m0 + TM <= m_lim && n0 + TN <= a.N) nt_epilogue_body<EPI, true, SC>(a, acc, s1, s2, m0, n0, m_lim, osc);
    else nt_epilogue_body<EPI, false, SC>(a, acc, s1, s2, m0, n0, m_lim, osc);
    if (a.part) {
        // combine the lane halves, then the two waves that share these columns, in a fixed order
        // red: [2 (wm)][2 (stat)][128 (col)]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s1[j] += __shfl_xor(s1[j], 32, 64);
            s2[j] += __shfl_xor(s2[j], 32, 64);
            if (h == 0) {
                const int col = wn * 64 + j * 32 + l31;
                red[(wm * 2 + 0) * TN + col] = s1[j];
                red[(wm * 2 + 1) * TN + col] = s2[j];
            }
        }
        __syncthreads();
        if (tid < TN && n0 + tid < a.N) {
            a.part[((long long)mt * 2 + 0) * a.N + n0 + tid] = red[0 * TN + tid] + red[2 * TN + tid];
            a.part[((long long)mt * 2 + 1) * a.N + n0 + tid] = red[1 * TN + tid] + red[3 * TN + tid];
        }
    }
}


// argument block of the split-bf16 convolution kernel (ign_clconv_x6.hip)
struct ConvX6Args {
    GemmNTArgs g;                 // C, M, N, bias, prologue, epilogue pointers, part, B3 (planes; Kp = k*Cp), mtiles = B*tps, ntiles
    long long sample_pitch;       // floats between samples of the input
    int rows_in;                  // valid input rows per sample (loads are clamped to it)
    int cin, cp, k;               // input channels, padded channels, taps
    int trows;                    // valid output rows per sample
    int tps;                      // m-tiles per sample
    int nprod;                    // 6: three-way bf16 split, six partial products (fp32 accuracy); 1: operands rounded to bf16;
                                  // 3: two-way fp16 split of power-of-two-scaled operands, three partial products (fp32 accuracy)
    const float* bound_a;         // nprod 3: device scalars, upper bounds of |A operand| (after the prologue) and of |weights|: the
    const float* bound_b;         //          operands are scaled by ign_pow2_scale(bound) before the split (null: 1)
};

// Power-of-two scale for the two-plane fp16 split: 2^e with 2^13 <= bound * 2^e < 2^14, so that the leading fp16 term of every
// element stays far from fp16's overflow (65504) and the second term of typical elements stays a NORMAL fp16 number
// (1 for a null pointer, zero, or a non-finite bound; the exponent is clamped to +-60 so that products of two scales stay finite).
__device__ __forceinline__ float ign_pow2_scale(const float* bound) {
    if (!bound) return 1.f;
    const float b = *bound;
    if (!(b > 0.f) || !(b < INFINITY)) return 1.f;
    int e;
    (void)frexpf(b, &e);                     // b = m * 2^e, m in [0.5, 1)
    e = 14 - e;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    return ldexpf(1.f, e);
}

inline int ign_vec_width(int c) { return (c % 4 == 0) ? 4 : (c % 2 == 0) ? 2 : 1; }

// host-side launchers shared across the translation units (a __global__ function can only be launched from its own TU)
int ign_clconv_launch_x6t(const ConvX6Args& a, int epi, int V, bool pro, hipStream_t s);     // ign_clconv_x6.hip
int ign_clconv_launch_wgrad_reduce(const float* part, float* dw_oik, int nsplit, int Co, int Ci, int k, hipStream_t s);   // _f32.hip

// Channels-last 1-D convolution as an implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), with the
// BatchNorm / ReLU glue of the FCN expert folded into the GEMM prologues and epilogues.
//
// Replaces IGN/model/FullyConvNet.py:31-50 (3 x [Conv1d -> BatchNorm1d -> ReLU]) and its autograd.
//
// Why a GEMM with no im2col: activations are (B, T, C) row-major -- the loader's own layout.  The im2col row of
// output position (b, t) is x[b, t : t+k, :], i.e. k*C CONTIGUOUS floats starting at x[(b*T + t)*C].  So
//     y[m][co] = sum_kk A[m][kk] * Wt[co][kk],   A[m][kk] = x[rowoff(m) + kk],   kk = j*C + ci,
// is an NT GEMM whose A rows overlap in memory (row pitch C, row length k*C); nothing is gathered or copied.
//   forward : A = previous activation (optionally relu(a_c*y + b_c) applied while staging: BatchNorm+ReLU of the
//             previous block never touch HBM), epilogue adds the bias and emits per-channel sum / sum-of-squares
//             partials for this block's BatchNorm;
//   dgrad   : the same kernel on the zero-padded output gradient with the tap-reversed weights; its epilogue applies
//             the ReLU mask of the layer below and emits that layer's BatchNorm-backward sums;
//   wgrad   : TN GEMM dW[co][kk] = sum_m dy[m][co] * A[m][kk], split over row ranges, fixed-order reduction.
// Tiles: 256-thread workgroups, 128x128 output tile, four waves in 2x2 each owning 64x64 = four 32x32 accumulators,
// 16-deep K chunks double-buffered in LDS (40 KB): global -> registers -> LDS, one barrier per chunk.
// The k index of a 32x32x2 step is permuted so that a lane's operand stream is 4 CONTIGUOUS floats (ds_read_b128):
// step (u,e) of a chunk takes k = 8u + 4h + e from lane half h, for A and B alike, so the dot product is unchanged.
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

enum { EPI_BIAS_STATS = 0, EPI_MASK_STATS = 1 };

struct GemmNTArgs {
    const float* A; RowMap am; int K;          // A[m][kk] = A[row_off(m) + kk]
    const float* Bt; int ldb;                  // Bt[n][kk]
    float* C; int M, N;                        // dense (M, N)
    const float* bias;                         // [N] or null
    const float* pro_a; const float* pro_b; int pro_c;    // prologue: A <- relu(pro_a[c]*A + pro_b[c]), c = kk % pro_c
    float* part;                               // (mtiles, 2, N) partial sums, or null
    // EPI_MASK_STATS: g = acc * [ea*y + eb > 0]; partials of g and g*(y - mean)*invstd
    const float* ey; const float* ea; const float* eb; const float* emean; const float* einv;
    int mtiles, ntiles;
    const unsigned short* B3; int Kp;          // split-bf16 kernels: Bt as 3 bf16 planes (3, N, Kp), Kp = K rounded up to 8
};

// ---- epilogue shared by the fp32 and the split-bf16 kernels.  Lane (l31, h) of accumulator (i, j) holds column
// n = n0 + wn*64 + j*32 + l31 and the 16 rows m0 + wm*64 + i*32 + acc_row16(r, h).  `red` is >= 512 floats of LDS that no
// wave still reads.  FULL = the whole 128x128 tile is inside the output: no per-element masks, and the 16 y values an
// accumulator needs (EPI_MASK_STATS) are fetched as 16 independent loads before any of them is used.
template <int EPI, bool FULL>
__device__ __forceinline__ void nt_epilogue_body(const GemmNTArgs& a, const f32x16 (&acc)[2][2], float (&s1)[2], float (&s2)[2],
                                                 int m0, int n0, int m_lim) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
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
            if (EPI == EPI_MASK_STATS) {
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
                float v = acc[i][j][r];
                if (EPI == EPI_BIAS_STATS) {
                    v += bv;
                    if (ok) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
                } else {
                    v = (fmaf(ea, yv[r], eb) > 0.f) ? v : 0.f;
                    if (ok) { s1[j] += v; s2[j] = fmaf(v, (yv[r] - em) * ei, s2[j]); }
                }
                if (ok) a.C[(long long)m * a.N + n] = v;
            }
        }
    }
}

template <int EPI>
__device__ __forceinline__ void nt_epilogue(const GemmNTArgs& a, const f32x16 (&acc)[2][2], float* red, int mt, int m0, int n0,
                                            int m_lim) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    if (m0 + TM <= m_lim && n0 + TN <= a.N) nt_epilogue_body<EPI, true>(a, acc, s1, s2, m0, n0, m_lim);
    else nt_epilogue_body<EPI, false>(a, acc, s1, s2, m0, n0, m_lim);
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

// ------------------------------------------------------------------------------------------------ NT GEMM
template <int V, bool PRO, int EPI>
__global__ void __launch_bounds__(256, 2) clconv_nt_kernel(const GemmNTArgs a) {
    constexpr int VPR = KC / V;                // vectors per staged row
    constexpr int RPP = 256 / VPR;             // rows per staging pass
    constexpr int NPASS = TM / RPP;            // passes per tile (8 / V)
    __shared__ __attribute__((aligned(16))) float smem[2][(TM + TN) * NT_PITCH];

    // XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2); give each XCD a contiguous run of
    // logical tiles so the n-tiles of one m-tile (same A rows) and neighbouring m-tiles (overlapping rows) meet there.
    const int nwg = a.mtiles * a.ntiles;
    int lid = blockIdx.x;
    {
        const int per = nwg / 8;
        if (lid < per * 8) lid = (lid & 7) * per + (lid >> 3);
    }
    const int mt = lid / a.ntiles, nt = lid - mt * a.ntiles;
    const int m0 = mt * TM, n0 = nt * TN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    // staging coordinates: vector q of rows (r + p*RPP)
    const int sq = tid % VPR, sr = tid / VPR;
    const float* arow[NPASS];
    const float* brow[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int m = min(m0 + sr + p * RPP, a.M - 1);        // tail rows shadow the last valid row (masked at the store)
        const int n = min(n0 + sr + p * RPP, a.N - 1);
        arow[p] = a.A + row_off(a.am, m) + sq * V;
        brow[p] = a.Bt + (long long)n * a.ldb + sq * V;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[NPASS][V], rb[NPASS][V];
    const int nchunk = (a.K + KC - 1) / KC;

    // The prologue (BatchNorm affine + ReLU of the block below) is applied when the registers are written to LDS, i.e.
    // AFTER the MFMAs of the current chunk: applying it at the load would make the wave wait for the load first.
    float pa[V], pb[V];
    bool pro_ok = false;
    auto gload = [&](int c) {
        const int kk = c * KC + sq * V;
        const bool ok = kk < a.K;               // V divides K: a vector is entirely inside or outside
        if (PRO) {
            pro_ok = ok;
            if (ok) {
                const int ch = kk % a.pro_c;
                vload<V>(pa, a.pro_a + ch);
                vload<V>(pb, a.pro_b + ch);
            }
        }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if (ok) {
                vload<V>(ra[p], arow[p] + c * KC);
                vload<V>(rb[p], brow[p] + c * KC);
            } else {
#pragma unroll
                for (int v = 0; v < V; ++v) { ra[p][v] = 0.f; rb[p][v] = 0.f; }
            }
        }
    };
    auto lstore = [&](int buf) {
        float* As = smem[buf];
        float* Bs = As + TM * NT_PITCH;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            if (PRO && pro_ok) {
#pragma unroll
                for (int v = 0; v < V; ++v) ra[p][v] = fmaxf(fmaf(pa[v], ra[p][v], pb[v]), 0.f);
            }
            vstore<V>(As + (sr + p * RPP) * NT_PITCH + sq * V, ra[p]);
            vstore<V>(Bs + (sr + p * RPP) * NT_PITCH + sq * V, rb[p]);
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        if (c + 1 < nchunk) gload(c + 1);
        const float* As = smem[buf] + (wm * 64 + l31) * NT_PITCH + 4 * h;
        const float* Bs = smem[buf] + TM * NT_PITCH + (wn * 64 + l31) * NT_PITCH + 4 * h;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float4 a0 = *reinterpret_cast<const float4*>(As + 8 * u);
            const float4 a1 = *reinterpret_cast<const float4*>(As + 32 * NT_PITCH + 8 * u);
            const float4 b0 = *reinterpret_cast<const float4*>(Bs + 8 * u);
            const float4 b1 = *reinterpret_cast<const float4*>(Bs + 32 * NT_PITCH + 8 * u);
#define IGN_STEP(e)                                   \
            acc[0][0] = MFMA32(a0.e, b0.e, acc[0][0]); \
            acc[0][1] = MFMA32(a0.e, b1.e, acc[0][1]); \
            acc[1][0] = MFMA32(a1.e, b0.e, acc[1][0]); \
            acc[1][1] = MFMA32(a1.e, b1.e, acc[1][1]);
            IGN_STEP(x) IGN_STEP(y) IGN_STEP(z) IGN_STEP(w)
#undef IGN_STEP
        }
        if (c + 1 < nchunk) lstore(buf ^ 1);
        __syncthreads();
    }

    nt_epilogue<EPI>(a, acc, smem[0], mt, m0, n0, a.M);
}

// ------------------------------------------------------------------------------------------------ NT GEMM, split bf16
// fp32-accurate product on the bf16 matrix cores.  gfx950 runs v_mfma_f32_32x32x2_f32 at the fp32 VECTOR rate
// (64 FLOP/clk/SIMD, 157 TFLOP/s) -- the real matrix throughput is behind the 16-bit inputs (1024 FLOP/clk/SIMD).
// Each fp32 operand is split exactly into three bf16 terms x = x0 + x1 + x2 (8 significant bits each, round-to-nearest;
// the residuals are exact fp32 subtractions) and the product keeps the six terms of weight >= 2^-16,
//     a*b ~= a0*b0 + (a0*b1 + a1*b0) + (a0*b2 + a1*b1 + a2*b0),        dropped: O(2^-24 |a b|),
// every partial product exact (8 x 8 bits) and accumulated in fp32 by v_mfma_f32_32x32x16_bf16: 6 MFMAs of 16x the
// rate = 2.7x the fp32-MFMA throughput at fp32 rounding-level error (measured against float64 in tests/test_gpu_fcn.py).
// The activation operand is split while it is staged (after the BatchNorm+ReLU prologue); the weights arrive pre-split
// (ign_clconv_pack_weights_x3).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int X6_PITCH = KC + 8;             // bf16 per staged row: 48 B, 12*i mod 64 dwords is conflict-free for b128 reads
constexpr int X6_PLANE = TM * X6_PITCH;      // bf16 per plane per stage

__device__ __forceinline__ void split3(float v, __bf16& x0, __bf16& x1, __bf16& x2) {
    x0 = (__bf16)v;
    float r = v - (float)x0;
    x1 = (__bf16)r;
    r -= (float)x1;
    x2 = (__bf16)r;
}

// ---- split-bf16 convolution with tap reuse.  The im2col rows of 128 consecutive output positions of one sample overlap:
// together they are the (128 + k - 1)-row span of the input.  Per 16-channel chunk that span is loaded, passed through the
// prologue and split ONCE, and all k taps run from it with the MFMA row operand shifted by one LDS row per tap -- the plain
// GEMM form above loads and splits every activation k times.  Weights: bf16 planes (3, N, k*Cp), Cp = channels rounded up to
// 16, so every (tap, chunk) block is 16-byte aligned.  One barrier per (chunk, tap) step of 24 MFMAs per wave; the weights
// of the next step and (on the last tap) the next span are prefetched into registers during the MFMAs.
struct ConvX6Args {
    GemmNTArgs g;                 // C, M, N, bias, prologue, epilogue pointers, part, B3 (planes; Kp = k*Cp), mtiles = B*tps, ntiles
    long long sample_pitch;       // floats between samples of the input
    int rows_in;                  // valid input rows per sample (loads are clamped to it)
    int cin, cp, k;               // input channels, padded channels, taps
    int trows;                    // valid output rows per sample
    int tps;                      // m-tiles per sample
};
constexpr int X6T_SPAN = TM + 15;                  // k <= 16
constexpr int X6T_APLANE = X6T_SPAN * X6_PITCH;
constexpr int X6T_ABUF = 3 * X6T_APLANE;
constexpr int X6T_BBUF = 3 * X6_PLANE;
constexpr size_t X6T_LDS_BYTES = (size_t)2 * (X6T_ABUF + X6T_BBUF) * sizeof(unsigned short);

template <int V, bool PRO, int EPI>
__global__ void __launch_bounds__(256, 2) clconv_x6t_kernel(const ConvX6Args ca) {
    const GemmNTArgs& a = ca.g;
    constexpr int VPR = KC / V;                               // vectors per span row
    constexpr int APASS = (X6T_SPAN * VPR + 255) / 256;       // staging passes of the span (V=4: 3, V=2: 5, V=1: 9)
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    __bf16* Abuf = reinterpret_cast<__bf16*>(smem16);
    __bf16* Bbuf = Abuf + 2 * X6T_ABUF;

    const int nwg = a.mtiles * a.ntiles;
    int lid = blockIdx.x;
    {
        const int per = nwg / 8;
        if (lid < per * 8) lid = (lid & 7) * per + (lid >> 3);
    }
    const int mt = lid / a.ntiles, nt = lid - mt * a.ntiles;
    const int bi = mt / ca.tps, tt = mt - bi * ca.tps;
    const int t0 = tt * TM, n0 = nt * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const int span = TM + ca.k - 1;
    const int nvec = span * VPR;

    const float* abase = a.A + (long long)bi * ca.sample_pitch;
    const int brw = tid >> 1, bh = tid & 1;
    const unsigned short* bsrc = a.B3 + (size_t)min(n0 + brw, a.N - 1) * a.Kp + 8 * bh;
    const size_t bplane = (size_t)a.N * a.Kp;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[APASS][V];
    float pa[V], pb[V];
    bool a_ok = false;
    // Weight prefetch registers: two sets of three planes (scalars, not arrays: arrays captured by the lambdas below end
    // up in scratch).  The loads for step s+2 are issued at the top of step s, so they have two steps to land.
    uint4 rb00, rb01, rb02, rb10, rb11, rb12;
    const int ncc = ca.cp / KC;
    const int nstep = ncc * ca.k;

    auto aload = [&](int cc) {
        const int ch = cc * KC + (tid % VPR) * V;          // 256 % VPR == 0: a thread keeps its channel offset in every pass
        a_ok = ch < ca.cin;                                  // V divides cin: a vector is entirely inside or outside
        if (PRO && a_ok) { vload<V>(pa, a.pro_a + ch); vload<V>(pb, a.pro_b + ch); }
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            const int idx = tid + p * 256;
            const int row = idx / VPR;
            if (a_ok && idx < nvec) vload<V>(ra[p], abase + (long long)min(t0 + row, ca.rows_in - 1) * ca.cin + ch);
            else {
#pragma unroll
                for (int v = 0; v < V; ++v) ra[p][v] = 0.f;
            }
        }
    };
    auto astore = [&](int buf) {
        __bf16* st = Abuf + buf * X6T_ABUF;
#pragma unroll
        for (int p = 0; p < APASS; ++p) {
            const int idx = tid + p * 256;
            if (idx < nvec) {
                const int row = idx / VPR, q = idx - row * VPR;
                __bf16 x0[V], x1[V], x2[V];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    float t = ra[p][v];
                    if (PRO && a_ok) t = fmaxf(fmaf(pa[v], t, pb[v]), 0.f);
                    split3(t, x0[v], x1[v], x2[v]);
                }
                __bf16* d = st + row * X6_PITCH + q * V;
#pragma unroll
                for (int v = 0; v < V; ++v) { d[v] = x0[v]; d[X6T_APLANE + v] = x1[v]; d[2 * X6T_APLANE + v] = x2[v]; }
            }
        }
    };
    // step index -> offset of its (tap, chunk) block in a weight row; steps past the end re-read the last block (the loads
    // stay unconditional so that the compiler's vmcnt bookkeeping is exact on every path)
    auto boff = [&](int st) {
        st = min(st, nstep - 1);
        const int cc = st / ca.k, j = st - cc * ca.k;
        return (size_t)j * ca.cp + cc * KC;
    };
    auto bstore = [&](int buf, const uint4& r0, const uint4& r1, const uint4& r2) {
        __bf16* st = Bbuf + buf * X6T_BBUF + brw * X6_PITCH + 8 * bh;
        *reinterpret_cast<uint4*>(st) = r0;
        *reinterpret_cast<uint4*>(st + X6_PLANE) = r1;
        *reinterpret_cast<uint4*>(st + 2 * X6_PLANE) = r2;
    };
#define IGN_BLOAD(r0, r1, r2, st)                                                       \
    do {                                                                                \
        const size_t off_ = boff(st);                                                   \
        r0 = *reinterpret_cast<const uint4*>(bsrc + off_);                              \
        r1 = *reinterpret_cast<const uint4*>(bsrc + bplane + off_);                     \
        r2 = *reinterpret_cast<const uint4*>(bsrc + 2 * bplane + off_);                 \
    } while (0)

    aload(0);
    IGN_BLOAD(rb00, rb01, rb02, 0);
    IGN_BLOAD(rb10, rb11, rb12, 1);
    astore(0);
    bstore(0, rb00, rb01, rb02);
    __syncthreads();

    int cc = 0, j = 0;
    auto body = [&](int step, bool odd) {
        // registers of parity (step & 1) held step `step` (already in LDS): reuse them for step + 2
        if (odd) IGN_BLOAD(rb10, rb11, rb12, step + 2);
        else IGN_BLOAD(rb00, rb01, rb02, step + 2);
        const bool next_span = cc + 1 < ncc;
        if (j == 0 && next_span) aload(cc + 1);             // k steps ahead of its use
        const __bf16* As = Abuf + (cc & 1) * X6T_ABUF + (wm * 64 + l31 + j) * X6_PITCH + 8 * h;
        const __bf16* Bs = Bbuf + (step & 1) * X6T_BBUF + (wn * 64 + l31) * X6_PITCH + 8 * h;
        bf16x8 af[2][3], bf[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                af[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * X6T_APLANE + i * 32 * X6_PITCH);
                bf[i][pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * X6_PLANE + i * 32 * X6_PITCH);
            }
#define IGN_X6(pa_, pb_)                                                                                   \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][pa_], bf[0][pb_], acc[0][0], 0, 0, 0);   \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][pa_], bf[1][pb_], acc[0][1], 0, 0, 0);   \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][pa_], bf[0][pb_], acc[1][0], 0, 0, 0);   \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][pa_], bf[1][pb_], acc[1][1], 0, 0, 0);
        IGN_X6(2, 0) IGN_X6(0, 2) IGN_X6(1, 1) IGN_X6(1, 0) IGN_X6(0, 1) IGN_X6(0, 0)
#undef IGN_X6
        // step + 1 goes to LDS from the OTHER register set (loaded during step - 1)
        if (odd) bstore((step + 1) & 1, rb00, rb01, rb02);
        else bstore((step + 1) & 1, rb10, rb11, rb12);
        if (j + 1 == ca.k) {
            if (next_span) astore((cc + 1) & 1);
            j = 0; ++cc;
        } else {
            ++j;
        }
        __syncthreads();
    };
    for (int step = 0; step < nstep; step += 2) {
        body(step, false);
        if (step + 1 < nstep) body(step + 1, true);
    }
#undef IGN_BLOAD
    const int m0 = bi * ca.trows + t0;
    nt_epilogue<EPI>(a, acc, reinterpret_cast<float*>(smem16), mt, m0, n0, bi * ca.trows + min(ca.trows, t0 + TM));
}

// Tap-major planes for the kernel above: Wt3[p][co][j*Cip + ci], Wd3[p][ci][jj*Cop + co] (zero in the padded channels)
__global__ void __launch_bounds__(256) pack_weights_x3t_kernel(const float* __restrict__ w, unsigned short* __restrict__ wt3,
                                                               unsigned short* __restrict__ wd3, int Co, int Ci, int k, int Cip,
                                                               int Cop) {
    const long long nf = (long long)Co * k * Cip, nd = wd3 ? (long long)Ci * k * Cop : 0;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    __bf16* dst;
    long long plane;
    if (i < nf) {
        const int ci = (int)(i % Cip);
        const long long t = i / Cip;
        const int j = (int)(t % k);
        const long long co = t / k;
        if (ci < Ci) v = w[(co * Ci + ci) * k + j];
        dst = reinterpret_cast<__bf16*>(wt3) + i;
        plane = nf;
    } else if (i < nf + nd) {
        const long long e = i - nf;
        const int co = (int)(e % Cop);
        const long long t = e / Cop;
        const int jj = (int)(t % k);
        const long long ci = t / k;
        if (co < Co) v = w[((long long)co * Ci + ci) * k + (k - 1 - jj)];
        dst = reinterpret_cast<__bf16*>(wd3) + e;
        plane = nd;
    } else {
        return;
    }
    __bf16 x0, x1, x2;
    split3(v, x0, x1, x2);
    dst[0] = x0; dst[plane] = x1; dst[2 * plane] = x2;
}

// ------------------------------------------------------------------------------------------------ TN GEMM (wgrad)
struct GemmTNArgs {
    const float* P; RowMap pm; int NP;         // P[m][co], co < NP (dense row of NP floats at row_off(pm, m))
    const float* Q; RowMap qm; int NQ;         // Q[m][kk] = Q[row_off(qm, m) + kk], kk < NQ
    const float* pro_a; const float* pro_b; int pro_c;    // Q <- relu(pro_a[c]*Q + pro_b[c]), c = kk % pro_c
    float* part;                               // (nsplit, NP, NQ) partial products
    int M, nsplit, ptiles, qtiles;
};

// Row-pair permutation: lane i of operand block b reads output index 2*i + b, so one ds_read_b64 feeds both 32-wide
// blocks of the wave tile; accumulator (i, j) row rho <-> co = 64*wm + 2*rho + i, lane column c <-> kk = 64*wn + 2*c + j.
template <int VQ, bool PRO>
__global__ void __launch_bounds__(256, 2) clconv_tn_kernel(const GemmTNArgs a) {
    constexpr int QVPR = 128 / VQ;             // Q vectors per staged row
    constexpr int QRPP = 256 / QVPR;           // Q rows per pass
    constexpr int QNPASS = KC / QRPP;          // (VQ=4: 2, VQ=2: 4, VQ=1: 8)
    __shared__ __attribute__((aligned(16))) float smem[2][2 * KC * TN_PITCH];

    const int tile = blockIdx.x;
    const int pt = tile / a.qtiles, qt = tile - pt * a.qtiles;
    const int p0 = pt * 128, q0 = qt * 128;
    const int split = blockIdx.y;
    const int rows_per = ((a.M + a.nsplit - 1) / a.nsplit + KC - 1) / KC * KC;
    const int m_begin = split * rows_per;
    const int m_end = min(a.M, m_begin + rows_per);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;

    // P staging: float4 c4 of rows r, r+8  (NP is a multiple of 4: checked by the launcher)
    const int pc = (tid & 31) * 4, pr = tid >> 5;
    const bool p_ok = p0 + pc < a.NP;
    // Q staging: vector qv of rows qr + i*QRPP
    const int qc = (tid % QVPR) * VQ, qr = tid / QVPR;
    const bool q_ok = q0 + qc < a.NQ;
    float qa[VQ], qb[VQ];
    if (PRO && q_ok) {
        const int ch = (q0 + qc) % a.pro_c;
        vload<VQ>(qa, a.pro_a + ch);
        vload<VQ>(qb, a.pro_b + ch);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float rp[2][4], rq[QNPASS][VQ];
    bool rq_ok[QNPASS];
    auto gload = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mb + pr + 8 * i;
            if (p_ok && m < m_end) vload<4>(rp[i], a.P + row_off(a.pm, m) + p0 + pc);
            else { rp[i][0] = rp[i][1] = rp[i][2] = rp[i][3] = 0.f; }
        }
#pragma unroll
        for (int i = 0; i < QNPASS; ++i) {
            const int m = mb + qr + QRPP * i;
            if (q_ok && m < m_end) {
                vload<VQ>(rq[i], a.Q + row_off(a.qm, m) + q0 + qc);
                rq_ok[i] = true;
            } else {
#pragma unroll
                for (int v = 0; v < VQ; ++v) rq[i][v] = 0.f;
                rq_ok[i] = false;
            }
        }
    };
    auto lstore = [&](int buf) {
        float* Ps = smem[buf];
        float* Qs = Ps + KC * TN_PITCH;
#pragma unroll
        for (int i = 0; i < 2; ++i) vstore<4>(Ps + (pr + 8 * i) * TN_PITCH + pc, rp[i]);
#pragma unroll
        for (int i = 0; i < QNPASS; ++i) {
            if (PRO && rq_ok[i]) {              // applied after the MFMAs of the current chunk (see the NT kernel)
#pragma unroll
                for (int v = 0; v < VQ; ++v) rq[i][v] = fmaxf(fmaf(qa[v], rq[i][v], qb[v]), 0.f);
            }
            vstore<VQ>(Qs + (qr + QRPP * i) * TN_PITCH + qc, rq[i]);
        }
    };

    if (m_begin < m_end) {
        gload(m_begin);
        lstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int mb = m_begin; mb < m_end; mb += KC, buf ^= 1) {
        if (mb + KC < m_end) gload(mb + KC);
        const float* Ps = smem[buf] + h * TN_PITCH + wm * 64 + 2 * l31;
        const float* Qs = smem[buf] + KC * TN_PITCH + h * TN_PITCH + wn * 64 + 2 * l31;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            const float2 pv = *reinterpret_cast<const float2*>(Ps + 2 * s * TN_PITCH);
            const float2 qv = *reinterpret_cast<const float2*>(Qs + 2 * s * TN_PITCH);
            acc[0][0] = MFMA32(pv.x, qv.x, acc[0][0]);
            acc[0][1] = MFMA32(pv.x, qv.y, acc[0][1]);
            acc[1][0] = MFMA32(pv.y, qv.x, acc[1][0]);
            acc[1][1] = MFMA32(pv.y, qv.y, acc[1][1]);
        }
        if (mb + KC < m_end) lstore(buf ^ 1);
        __syncthreads();
    }

    float* out = a.part + (long long)split * a.NP * a.NQ;
    const int kk = q0 + wn * 64 + 2 * l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = p0 + wm * 64 + 2 * acc_row16(r, h) + i;
            if (co < a.NP) {
                float* o = out + (long long)co * a.NQ + kk;
                if (!(a.NQ & 1) && kk + 1 < a.NQ) {
                    *reinterpret_cast<float2*>(o) = make_float2(acc[i][0][r], acc[i][1][r]);
                } else {                       // odd row pitch: the pair is not 8-byte aligned
                    if (kk < a.NQ) o[0] = acc[i][0][r];
                    if (kk + 1 < a.NQ) o[1] = acc[i][1][r];
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ weight gradient, split bf16
// dW[co][j][ci] = sum_{b,t} dy[b,t,co] * in[b,t+j,ci] on the bf16 matrix cores (same six-term split as above).
// The reduction index is the ROW index of both operands, so the MFMA needs each operand column-major; the operands are
// staged row-major (as they arrive: coalesced loads, one split per element) and read with gfx950's transposing LDS read
// ds_read_b64_tr_b16, which hands lane i the 4 consecutive rows of column i.  Because rows stay rows in LDS, tap j is
// again just a row offset: one staged (16*NR + k - 1)-row span of the input serves all k taps, and the dy fragments are
// read once per unit and reused by every tap.  Workgroup tile: 64 co x 64 ci x k taps (wave: 32 x 32 x k taps = k
// accumulators), reduction over units of 16*NR output rows that never straddle a sample, split over unit ranges.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int WG_PITCH = 64 + 32;             // bf16 per staged row: 192 B; four rows' 32-dword windows tile the 64 banks

struct WgradX6Args {
    const float* dy; long long dy_sample_pitch; int dy_row0;     // dy[b][dy_row0 + t][co], row pitch Co
    const float* x; long long x_sample_pitch;                      // in[b][t][ci], row pitch Ci
    const float* pro_a; const float* pro_b;                        // in = relu(pro_a*x + pro_b) (or raw x)
    float* part;                                                   // (nsplit, Co, k*Ci)
    int B, Tin, Tout, Ci, Co, k;
    int cps;                                                       // units per sample
    int nunits, nsplit, citiles;
};

__device__ __forceinline__ bf16x8 lds_tr8(const __bf16* p0, const __bf16* p1) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int KT, int NR, int VX, bool PRO>
__global__ void __launch_bounds__(256, 2) clconv_wgrad_x6_kernel(const WgradX6Args a) {
    constexpr int RU = 16 * NR;                     // output rows per unit
    constexpr int SPAN = RU + KT - 1;               // input rows per unit
    constexpr int PPLANE = RU * WG_PITCH, QPLANE = SPAN * WG_PITCH;
    constexpr int STAGE = 3 * (PPLANE + QPLANE);
    constexpr int XVPR = 64 / VX;                   // input vectors per row
    constexpr int XPASS = (SPAN * XVPR + 255) / 256;
    constexpr int DPASS = NR;                       // dy: RU rows x 16 float4 = 256 * NR vectors
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    __bf16* smem = reinterpret_cast<__bf16*>(smem16);

    const int tile = blockIdx.x;
    const int cot = tile / a.citiles, cit = tile - cot * a.citiles;
    const int co0 = cot * 64, ci0 = cit * 64;
    const int split = blockIdx.y;
    const int per = (a.nunits + a.nsplit - 1) / a.nsplit;
    const int u_begin = split * per, u_end = min(a.nunits, u_begin + per);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wco = wave & 1, wci = wave >> 1;

    // staging coordinates
    const int dr = tid >> 4, dc = (tid & 15) * 4;            // dy: row dr (+16 per pass), 4 channels from dc
    const bool d_ok = co0 + dc < a.Co;                        // Co % 4 == 0
    const int xq = (tid % XVPR) * VX;                         // input: channel offset inside the tile (256 % XVPR == 0)
    const bool x_ok = ci0 + xq < a.Ci;                        // VX divides Ci
    float pa[VX], pb[VX];
    if (PRO && x_ok) { vload<VX>(pa, a.pro_a + ci0 + xq); vload<VX>(pb, a.pro_b + ci0 + xq); }

    f32x16 acc[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    float rd[DPASS][4], rx[XPASS][VX];
    auto gload = [&](int u) {
        const int b = u / a.cps, t0 = (u - b * a.cps) * RU;
        const float* dyb = a.dy + (long long)b * a.dy_sample_pitch + (long long)a.dy_row0 * a.Co + co0 + dc;
        const float* xb = a.x + (long long)b * a.x_sample_pitch + ci0 + xq;
#pragma unroll
        for (int p = 0; p < DPASS; ++p) {
            const int t = t0 + dr + 16 * p;
            if (d_ok && t < a.Tout) vload<4>(rd[p], dyb + (long long)t * a.Co);
            else { rd[p][0] = rd[p][1] = rd[p][2] = rd[p][3] = 0.f; }
        }
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
            const int row = (tid + p * 256) / XVPR;
            if (x_ok && row < SPAN) vload<VX>(rx[p], xb + (long long)min(t0 + row, a.Tin - 1) * a.Ci);
            else {
#pragma unroll
                for (int v = 0; v < VX; ++v) rx[p][v] = 0.f;
            }
        }
    };
    auto lstore = [&](int buf) {
        __bf16* P = smem + buf * STAGE;
        __bf16* Q = P + 3 * PPLANE;
#pragma unroll
        for (int p = 0; p < DPASS; ++p) {
            __bf16* d = P + (dr + 16 * p) * WG_PITCH + dc;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                __bf16 x0, x1, x2;
                split3(rd[p][v], x0, x1, x2);
                d[v] = x0; d[PPLANE + v] = x1; d[2 * PPLANE + v] = x2;
            }
        }
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
            const int row = (tid + p * 256) / XVPR;
            if (row < SPAN) {
                __bf16* d = Q + row * WG_PITCH + xq;
#pragma unroll
                for (int v = 0; v < VX; ++v) {
                    float t = rx[p][v];
                    if (PRO && x_ok) t = fmaxf(fmaf(pa[v], t, pb[v]), 0.f);
                    __bf16 x0, x1, x2;
                    split3(t, x0, x1, x2);
                    d[v] = x0; d[QPLANE + v] = x1; d[2 * QPLANE + v] = x2;
                }
            }
        }
    };

    // transposed-read addresses: 16-lane group G = lane>>4 reads the 4-row x 16-column block whose rows are supplied by
    // lanes 4q+p (row q, columns 4p..4p+3); lane u of the group receives column u.  For the 32x32x16 operand lane (i, g)
    // needs rows 8g..8g+7 of column i: two reads (rows 8g..8g+3 and 8g+4..8g+7) of the block at columns 16*(G&1).
    const int u16 = lane & 15, q4 = u16 >> 2, p4 = u16 & 3, G1 = (lane >> 4) & 1;
    const int row_lo = 8 * h + q4;
    const int acol = wco * 32 + 16 * G1 + 4 * p4;
    const int bcol = wci * 32 + 16 * G1 + 4 * p4;

    if (u_begin < u_end) {
        gload(u_begin);
        lstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int u = u_begin; u < u_end; ++u, buf ^= 1) {
        if (u + 1 < u_end) gload(u + 1);
        const __bf16* P = smem + buf * STAGE;
        const __bf16* Q = P + 3 * PPLANE;
#pragma unroll
        for (int s = 0; s < NR; ++s) {
            bf16x8 af[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const __bf16* base = P + pl * PPLANE + (16 * s + row_lo) * WG_PITCH + acol;
                af[pl] = lds_tr8(base, base + 4 * WG_PITCH);
            }
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                bf16x8 bf[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const __bf16* base = Q + pl * QPLANE + (16 * s + row_lo + j) * WG_PITCH + bcol;
                    bf[pl] = lds_tr8(base, base + 4 * WG_PITCH);
                }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[j], 0, 0, 0);
            }
        }
        if (u + 1 < u_end) lstore(buf ^ 1);
        __syncthreads();
    }

    float* out = a.part + (long long)split * a.Co * a.k * a.Ci;
    const int ci = ci0 + wci * 32 + l31;
    if (ci < a.Ci) {
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wco * 32 + acc_row16(r, h);
                if (co < a.Co) out[((long long)co * a.k + j) * a.Ci + ci] = acc[j][r];
            }
    }
}

// dW[co][ci][j] = sum_s part[s][co][j*Ci + ci]   (s ascending: bitwise reproducible); torch (Co, Ci, k) layout
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nsplit,
                                                           int Co, int Ci, int k) {
    const long long n = (long long)Co * Ci * k;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // index in (Co, k, Ci) order: coalesced reads
    if (i >= n) return;
    float s = 0.f;
    for (int p = 0; p < nsplit; ++p) s += part[(long long)p * n + i];
    const int ci = (int)(i % Ci);
    const long long t = i / Ci;
    const int j = (int)(t % k);
    const long long co = t / k;
    dw[(co * Ci + ci) * k + j] = s;
}

// Wt[co][j*Ci + ci] = W[co][ci][j];   Wd[ci][jj*Co + co] = W[co][ci][k-1-jj]
__global__ void __launch_bounds__(256) pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                           float* __restrict__ wd, int Co, int Ci, int k) {
    const long long n = (long long)Co * Ci * k;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int j = (int)(i % k);
    const long long t = i / k;
    const int ci = (int)(t % Ci);
    const long long co = t / Ci;
    const float v = w[i];
    wt[(co * k + j) * Ci + ci] = v;
    if (wd) wd[((long long)ci * k + (k - 1 - j)) * Co + co] = v;
}

// ------------------------------------------------------------------------------------------------ BatchNorm glue
// Sum of the per-tile partials (nparts, 2, C) for 32 channels per block: 32 slices of the partials are summed in
// parallel (ascending inside a slice), then the 32 slice sums are combined in fixed order in double.
__device__ __forceinline__ void bn_sum_partials(const float* __restrict__ part, int nparts, int C, double* s_out, double* q_out) {
    __shared__ double sh[2][32][33];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        const int per = (nparts + 31) / 32;
        const int i0 = sl * per, i1 = min(nparts, i0 + per);
        float fs = 0.f, fq = 0.f;              // <= 64 addends of like magnitude per slice in fp32, slices combined in double
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
        for (int k = 0; k < 32; ++k) { s += sh[0][k][cl]; q += sh[1][k][cl]; }
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
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) != 0 || c >= C) return;
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
    const int c = blockIdx.x * 32 + (threadIdx.x & 31);
    if ((threadIdx.x >> 5) != 0 || c >= C) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
}

// Global average pool of relu(a*y + b) over time:  pooled[b][c] = (1/T) sum_t max(a_c*y[b,t,c] + b_c, 0).
// One block per sample; thread <-> (row lane, 4 channels); the row lanes are combined through LDS in fixed order.
__global__ void __launch_bounds__(256) bn_relu_pool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ a,
                                                               const float* __restrict__ b, float* __restrict__ pooled, int T,
                                                               int C) {
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
        pooled[(size_t)blockIdx.x * C + c] = u / (float)T;
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
                                                           int pad, long long rows_padded, float invR, int training) {
    const int c4 = C / 4, rif = 256 / c4;
    const int col = (threadIdx.x % c4) * 4, rofs = threadIdx.x / c4;
    if (rofs >= rif) return;
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
            for (int q = 0; q < 4; ++q) out[q] = fmaf(ca[q], gg[q], fmaf(c1[q], yy[q], c0[q]));
        }
        vstore<4>(dyp + (size_t)row * C + col, out);
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
static int vec_width(int c) { return (c % 4 == 0) ? 4 : (c % 2 == 0) ? 2 : 1; }

template <int EPI>
static int launch_nt(const GemmNTArgs& a, int V, bool pro, hipStream_t s) {
    const dim3 grid((unsigned)(a.mtiles * a.ntiles)), block(256);
#define IGN_NT(VV, PP) hipLaunchKernelGGL((clconv_nt_kernel<VV, PP, EPI>), grid, block, 0, s, a)
    if (V == 4) { if (pro) IGN_NT(4, true); else IGN_NT(4, false); }
    else if (V == 2) { if (pro) IGN_NT(2, true); else IGN_NT(2, false); }
    else { if (pro) IGN_NT(1, true); else IGN_NT(1, false); }
#undef IGN_NT
    return ign_check_launch("clconv_nt_kernel");
}

template <int EPI>
static int launch_x6t(const ConvX6Args& a, int V, bool pro, hipStream_t s) {
    const dim3 grid((unsigned)(a.g.mtiles * a.g.ntiles)), block(256);
#define IGN_X6T(VV, PP)                                                                                                      \
    do {                                                                                                                     \
        static bool once = false;                                                                                            \
        if (!once) {                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&clconv_x6t_kernel<VV, PP, EPI>),                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6T_LDS_BYTES);                       \
            once = true;                                                                                                     \
        }                                                                                                                    \
        hipLaunchKernelGGL((clconv_x6t_kernel<VV, PP, EPI>), grid, block, X6T_LDS_BYTES, s, a);                               \
    } while (0)
    if (V == 4) { if (pro) IGN_X6T(4, true); else IGN_X6T(4, false); }
    else if (V == 2) { if (pro) IGN_X6T(2, true); else IGN_X6T(2, false); }
    else { if (pro) IGN_X6T(1, true); else IGN_X6T(1, false); }
#undef IGN_X6T
    return ign_check_launch("clconv_x6t_kernel");
}

extern "C" int ign_clconv_kpad(int C) { return (C + 15) / 16 * 16; }
extern "C" long long ign_clconv_x6_mtiles(int B, int rows) { return (long long)B * ((rows + TM - 1) / TM); }

extern "C" int ign_clconv_pack_weights_x3(const float* w_oik, void* wt3_fwd, void* wt3_dgrad, int Co, int Ci, int k, void* stream) {
    if (!w_oik || !wt3_fwd || Co <= 0 || Ci <= 0 || k <= 0) {
        ign_set_error("ign_clconv_pack_weights_x3: bad argument (Co=%d Ci=%d k=%d)", Co, Ci, k);
        return IGN_E_ARG;
    }
    const int Cip = ign_clconv_kpad(Ci), Cop = ign_clconv_kpad(Co);
    const long long n = (long long)Co * k * Cip + (wt3_dgrad ? (long long)Ci * k * Cop : 0);
    hipLaunchKernelGGL(pack_weights_x3t_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oik,
                       (unsigned short*)wt3_fwd, (unsigned short*)wt3_dgrad, Co, Ci, k, Cip, Cop);
    return ign_check_launch("pack_weights_x3t_kernel");
}

extern "C" long long ign_clconv_mtiles(long long M) { return (M + TM - 1) / TM; }

extern "C" int ign_clconv_pack_weights(const float* w_oik, float* wt_fwd, float* wt_dgrad, int Co, int Ci, int k, void* stream) {
    if (!w_oik || !wt_fwd || Co <= 0 || Ci <= 0 || k <= 0) {
        ign_set_error("ign_clconv_pack_weights: bad argument (Co=%d Ci=%d k=%d)", Co, Ci, k);
        return IGN_E_ARG;
    }
    const long long n = (long long)Co * Ci * k;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oik, wt_fwd,
                       wt_dgrad, Co, Ci, k);
    return ign_check_launch("pack_weights_kernel");
}

static int clconv_fwd_impl(const char* who, bool x6, const float* x, const void* wt, const float* bias, const float* pro_a,
                           const float* pro_b, float* y, float* stat_part, int B, int Tin, int Ci, int Co, int k, void* stream) {
    const int Tout = Tin - k + 1;
    if (!x || !wt || !y || B <= 0 || Ci <= 0 || Co <= 0 || k <= 0 || Tout <= 0 || ((pro_a == nullptr) != (pro_b == nullptr))) {
        ign_set_error("%s: bad argument (B=%d Tin=%d Ci=%d Co=%d k=%d)", who, B, Tin, Ci, Co, k);
        return IGN_E_ARG;
    }
    const long long M = (long long)B * Tout;
    if (M > 0x7fffffffLL / 2) { ign_set_error("%s: B*Tout = %lld rows exceed the 2^30 row index space", who, M); return IGN_E_TOOBIG; }
    GemmNTArgs a{};
    a.A = x; a.am = RowMap{Tout, 0, Ci, (long long)Tin * Ci}; a.K = k * Ci;
    a.Bt = x6 ? nullptr : (const float*)wt; a.ldb = k * Ci; a.C = y; a.M = (int)M; a.N = Co; a.bias = bias;
    a.B3 = x6 ? (const unsigned short*)wt : nullptr; a.Kp = ign_clconv_kpad(k * Ci);
    a.pro_a = pro_a; a.pro_b = pro_b; a.pro_c = Ci; a.part = stat_part;
    a.mtiles = (int)((M + TM - 1) / TM); a.ntiles = (Co + TN - 1) / TN;
    IgnScopedTimer tm("clconv_fwd", (hipStream_t)stream);
    if (x6) {
        if (k > 16) { ign_set_error("%s: k=%d > 16 taps", who, k); return IGN_E_UNSUP; }
        ConvX6Args c{};
        c.g = a;
        c.cin = Ci; c.cp = ign_clconv_kpad(Ci); c.k = k; c.g.Kp = k * c.cp;
        c.sample_pitch = (long long)Tin * Ci; c.rows_in = Tin; c.trows = Tout; c.tps = (Tout + TM - 1) / TM;
        c.g.mtiles = B * c.tps;
        return launch_x6t<EPI_BIAS_STATS>(c, vec_width(Ci), pro_a != nullptr, (hipStream_t)stream);
    }
    return launch_nt<EPI_BIAS_STATS>(a, vec_width(Ci), pro_a != nullptr, (hipStream_t)stream);
}

extern "C" int ign_clconv_fwd(const float* x, const float* wt, const float* bias, const float* pro_a, const float* pro_b,
                              float* y, float* stat_part, int B, int Tin, int Ci, int Co, int k, void* stream) {
    return clconv_fwd_impl("ign_clconv_fwd", false, x, wt, bias, pro_a, pro_b, y, stat_part, B, Tin, Ci, Co, k, stream);
}

extern "C" int ign_clconv_fwd_x6(const float* x, const void* wt3, const float* bias, const float* pro_a, const float* pro_b,
                                 float* y, float* stat_part, int B, int Tin, int Ci, int Co, int k, void* stream) {
    return clconv_fwd_impl("ign_clconv_fwd_x6", true, x, wt3, bias, pro_a, pro_b, y, stat_part, B, Tin, Ci, Co, k, stream);
}

static int clconv_dgrad_impl(const char* who, bool x6, const float* dyp, const void* wt_dgrad, const float* y_in, const float* a_in,
                             const float* b_in, const float* mean_in, const float* invstd_in, float* g_in, float* stat_part, int B,
                             int Tin, int Ci, int Co, int k, void* stream) {
    const int Tout = Tin - k + 1;
    if (!dyp || !wt_dgrad || !y_in || !a_in || !b_in || !mean_in || !invstd_in || !g_in || B <= 0 || Ci <= 0 || Co <= 0 || k <= 0 ||
        Tout <= 0) {
        ign_set_error("%s: bad argument (B=%d Tin=%d Ci=%d Co=%d k=%d)", who, B, Tin, Ci, Co, k);
        return IGN_E_ARG;
    }
    const long long M = (long long)B * Tin;
    if (M > 0x7fffffffLL / 2) { ign_set_error("%s: B*Tin = %lld rows exceed the 2^30 row index space", who, M); return IGN_E_TOOBIG; }
    GemmNTArgs a{};
    // logical row (b, t) reads padded rows t .. t+k-1 of sample b: dz[b,t,ci] = sum_{jj,co} dyp[b,t+jj,co] W[co,ci,k-1-jj]
    a.A = dyp; a.am = RowMap{Tin, 0, Co, (long long)(Tout + 2 * (k - 1)) * Co}; a.K = k * Co;
    a.Bt = x6 ? nullptr : (const float*)wt_dgrad; a.ldb = k * Co; a.C = g_in; a.M = (int)M; a.N = Ci;
    a.B3 = x6 ? (const unsigned short*)wt_dgrad : nullptr; a.Kp = ign_clconv_kpad(k * Co);
    a.part = stat_part; a.ey = y_in; a.ea = a_in; a.eb = b_in; a.emean = mean_in; a.einv = invstd_in;
    a.mtiles = (int)((M + TM - 1) / TM); a.ntiles = (Ci + TN - 1) / TN;
    IgnScopedTimer tm("clconv_dgrad", (hipStream_t)stream);
    if (x6) {
        if (k > 16) { ign_set_error("%s: k=%d > 16 taps", who, k); return IGN_E_UNSUP; }
        ConvX6Args c{};
        c.g = a;
        c.cin = Co; c.cp = ign_clconv_kpad(Co); c.k = k; c.g.Kp = k * c.cp;
        c.sample_pitch = (long long)(Tout + 2 * (k - 1)) * Co; c.rows_in = Tout + 2 * (k - 1); c.trows = Tin; c.tps = (Tin + TM - 1) / TM;
        c.g.mtiles = B * c.tps;
        return launch_x6t<EPI_MASK_STATS>(c, vec_width(Co), false, (hipStream_t)stream);
    }
    return launch_nt<EPI_MASK_STATS>(a, vec_width(Co), false, (hipStream_t)stream);
}

extern "C" int ign_clconv_dgrad(const float* dyp, const float* wt_dgrad, const float* y_in, const float* a_in, const float* b_in,
                                const float* mean_in, const float* invstd_in, float* g_in, float* stat_part, int B, int Tin,
                                int Ci, int Co, int k, void* stream) {
    return clconv_dgrad_impl("ign_clconv_dgrad", false, dyp, wt_dgrad, y_in, a_in, b_in, mean_in, invstd_in, g_in, stat_part, B, Tin,
                             Ci, Co, k, stream);
}

extern "C" int ign_clconv_dgrad_x6(const float* dyp, const void* wt3_dgrad, const float* y_in, const float* a_in, const float* b_in,
                                   const float* mean_in, const float* invstd_in, float* g_in, float* stat_part, int B, int Tin,
                                   int Ci, int Co, int k, void* stream) {
    return clconv_dgrad_impl("ign_clconv_dgrad_x6", true, dyp, wt3_dgrad, y_in, a_in, b_in, mean_in, invstd_in, g_in, stat_part, B,
                             Tin, Ci, Co, k, stream);
}

static int wgrad_splits(long long M, int tiles) {
    long long s = (1024 + tiles - 1) / tiles;             // ~4 workgroups per CU in flight
    const long long max_s = (M + 8 * KC - 1) / (8 * KC);  // at least 8 chunks per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return (int)s;
}

extern "C" size_t ign_clconv_wgrad_workspace_bytes(int B, int Tin, int Ci, int Co, int k) {
    const int Tout = Tin - k + 1;
    if (B <= 0 || Tout <= 0 || Ci <= 0 || Co <= 0) return 0;
    const int tiles = ((Co + 127) / 128) * ((k * Ci + 127) / 128);
    return (size_t)wgrad_splits((long long)B * Tout, tiles) * Co * k * Ci * sizeof(float);
}

extern "C" int ign_clconv_wgrad(const float* dyp, int dy_pad, const float* x, const float* pro_a, const float* pro_b,
                                float* dw_oik, void* workspace, int B, int Tin, int Ci, int Co, int k, void* stream) {
    static const char* who = "ign_clconv_wgrad";
    const int Tout = Tin - k + 1;
    if (!dyp || !x || !dw_oik || !workspace || B <= 0 || Ci <= 0 || Co <= 0 || k <= 0 || Tout <= 0 || dy_pad < 0 ||
        ((pro_a == nullptr) != (pro_b == nullptr))) {
        ign_set_error("%s: bad argument (B=%d Tin=%d Ci=%d Co=%d k=%d pad=%d)", who, B, Tin, Ci, Co, k, dy_pad);
        return IGN_E_ARG;
    }
    if (Co % 4) { ign_set_error("%s: Co=%d must be a multiple of 4", who, Co); return IGN_E_UNSUP; }
    const long long M = (long long)B * Tout;
    if (M > 0x7fffffffLL / 2) { ign_set_error("%s: B*Tout = %lld rows exceed the 2^30 row index space", who, M); return IGN_E_TOOBIG; }
    hipStream_t s = (hipStream_t)stream;
    GemmTNArgs a{};
    a.P = dyp; a.pm = RowMap{Tout, dy_pad, Co, (long long)(Tout + 2 * dy_pad) * Co}; a.NP = Co;
    a.Q = x; a.qm = RowMap{Tout, 0, Ci, (long long)Tin * Ci}; a.NQ = k * Ci;
    a.pro_a = pro_a; a.pro_b = pro_b; a.pro_c = Ci;
    a.part = (float*)workspace; a.M = (int)M;
    a.ptiles = (Co + 127) / 128; a.qtiles = (k * Ci + 127) / 128;
    a.nsplit = wgrad_splits(M, a.ptiles * a.qtiles);
    const dim3 grid((unsigned)(a.ptiles * a.qtiles), (unsigned)a.nsplit), block(256);
    const int V = vec_width(Ci);
    const bool pro = pro_a != nullptr;
    {
        IgnScopedTimer tm("clconv_wgrad", s);
#define IGN_TN(VV, PP) hipLaunchKernelGGL((clconv_tn_kernel<VV, PP>), grid, block, 0, s, a)
        if (V == 4) { if (pro) IGN_TN(4, true); else IGN_TN(4, false); }
        else if (V == 2) { if (pro) IGN_TN(2, true); else IGN_TN(2, false); }
        else { if (pro) IGN_TN(1, true); else IGN_TN(1, false); }
#undef IGN_TN
    }
    int rc;
    if ((rc = ign_check_launch("clconv_tn_kernel"))) return rc;
    const long long n = (long long)Co * Ci * k;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)workspace, dw_oik,
                       a.nsplit, Co, Ci, k);
    return ign_check_launch("wgrad_reduce_kernel");
}

template <int KT, int NR>
static int launch_wgrad_x6(const WgradX6Args& a, int V, bool pro, dim3 grid, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * 3 * ((16 * NR) + (16 * NR + KT - 1)) * WG_PITCH * sizeof(unsigned short);
#define IGN_WG(VV, PP)                                                                                                       \
    do {                                                                                                                     \
        static bool once = false;                                                                                            \
        if (!once) {                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&clconv_wgrad_x6_kernel<KT, NR, VV, PP>),                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
            once = true;                                                                                                     \
        }                                                                                                                    \
        hipLaunchKernelGGL((clconv_wgrad_x6_kernel<KT, NR, VV, PP>), grid, dim3(256), lds, s, a);                             \
    } while (0)
    if (V == 4) { if (pro) IGN_WG(4, true); else IGN_WG(4, false); }
    else if (V == 2) { if (pro) IGN_WG(2, true); else IGN_WG(2, false); }
    else { if (pro) IGN_WG(1, true); else IGN_WG(1, false); }
#undef IGN_WG
    return ign_check_launch("clconv_wgrad_x6_kernel");
}

static int wgrad_x6_rows_per_unit(int k) { return (k == 8 || k == 5) ? 16 : (k == 3 || k == 2) ? 32 : 0; }
static int wgrad_x6_splits(int nunits, int tiles) {
    int s = (512 + tiles - 1) / tiles;                        // ~2 workgroups per CU
    const int max_s = (nunits + 15) / 16;                     // at least 16 units per split
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}

extern "C" size_t ign_clconv_wgrad_x6_workspace_bytes(int B, int Tin, int Ci, int Co, int k) {
    const int Tout = Tin - k + 1, ru = wgrad_x6_rows_per_unit(k);
    if (B <= 0 || Tout <= 0 || Ci <= 0 || Co <= 0 || !ru) return 0;
    const int tiles = ((Co + 63) / 64) * ((Ci + 63) / 64);
    const int nunits = B * ((Tout + ru - 1) / ru);
    return (size_t)wgrad_x6_splits(nunits, tiles) * Co * k * Ci * sizeof(float);
}

extern "C" int ign_clconv_wgrad_x6(const float* dyp, int dy_pad, const float* x, const float* pro_a, const float* pro_b,
                                   float* dw_oik, void* workspace, int B, int Tin, int Ci, int Co, int k, void* stream) {
    static const char* who = "ign_clconv_wgrad_x6";
    const int Tout = Tin - k + 1, ru = wgrad_x6_rows_per_unit(k);
    if (!dyp || !x || !dw_oik || !workspace || B <= 0 || Ci <= 0 || Co <= 0 || k <= 0 || Tout <= 0 || dy_pad < 0 ||
        ((pro_a == nullptr) != (pro_b == nullptr))) {
        ign_set_error("%s: bad argument (B=%d Tin=%d Ci=%d Co=%d k=%d pad=%d)", who, B, Tin, Ci, Co, k, dy_pad);
        return IGN_E_ARG;
    }
    if (Co % 4 || !ru) { ign_set_error("%s: needs Co %% 4 == 0 and k in {2,3,5,8} (Co=%d k=%d)", who, Co, k); return IGN_E_UNSUP; }
    hipStream_t s = (hipStream_t)stream;
    WgradX6Args a{};
    a.dy = dyp; a.dy_sample_pitch = (long long)(Tout + 2 * dy_pad) * Co; a.dy_row0 = dy_pad;
    a.x = x; a.x_sample_pitch = (long long)Tin * Ci; a.pro_a = pro_a; a.pro_b = pro_b;
    a.part = (float*)workspace; a.B = B; a.Tin = Tin; a.Tout = Tout; a.Ci = Ci; a.Co = Co; a.k = k;
    a.cps = (Tout + ru - 1) / ru; a.nunits = B * a.cps;
    a.citiles = (Ci + 63) / 64;
    const int tiles = ((Co + 63) / 64) * a.citiles;
    a.nsplit = wgrad_x6_splits(a.nunits, tiles);
    const dim3 grid((unsigned)tiles, (unsigned)a.nsplit);
    const int V = vec_width(Ci);
    const bool pro = pro_a != nullptr;
    int rc;
    {
        IgnScopedTimer tm("clconv_wgrad", s);
        if (k == 8) rc = launch_wgrad_x6<8, 1>(a, V, pro, grid, s);
        else if (k == 5) rc = launch_wgrad_x6<5, 1>(a, V, pro, grid, s);
        else if (k == 3) rc = launch_wgrad_x6<3, 2>(a, V, pro, grid, s);
        else rc = launch_wgrad_x6<2, 2>(a, V, pro, grid, s);
    }
    if (rc) return rc;
    const long long n = (long long)Co * Ci * k;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)workspace, dw_oik,
                       a.nsplit, Co, Ci, k);
    return ign_check_launch("wgrad_reduce_kernel");
}

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
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, part, nparts, R, C, gamma, beta,
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
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 31) / 32), dim3(1024), 0, (hipStream_t)stream, part, nparts, C, dbeta, dgamma);
    return ign_check_launch("bn_finalize_bwd_kernel");
}

extern "C" int ign_bn_relu_pool_fwd(const float* y, const float* a, const float* b, float* pooled, int B, int T, int C,
                                    void* stream) {
    int rc;
    if ((rc = bn_check("ign_bn_relu_pool_fwd", (long long)B * T, C))) return rc;
    if (!y || !a || !b || !pooled) { ign_set_error("ign_bn_relu_pool_fwd: null pointer"); return IGN_E_ARG; }
    const int rif = 256 / (C / 4);
    IgnScopedTimer tm("bn_relu_pool_fwd", (hipStream_t)stream);
    hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(B), dim3(256), (size_t)rif * C * 4, (hipStream_t)stream, y, a, b, pooled, T, C);
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
    int rc;
    if ((rc = bn_check("ign_bn_bwd_apply", (long long)B * T, C))) return rc;
    if (!g || !a || !dyp || pad < 0 || (training && (!y || !mean || !invstd || !dbeta || !dgamma))) {
        ign_set_error("ign_bn_bwd_apply: null pointer / negative pad");
        return IGN_E_ARG;
    }
    const long long rows = (long long)B * (T + 2 * pad);
    IgnScopedTimer tm("bn_bwd_apply", (hipStream_t)stream);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((rows + APPLY_ROWS - 1) / APPLY_ROWS)), dim3(256), 0, (hipStream_t)stream,
                       g, y, a, mean, invstd, dbeta, dgamma, dyp, T, C, pad, rows, 1.0f / (float)((long long)B * T), training);
    return ign_check_launch("bn_bwd_apply_kernel");
}

// Split-bf16 ("x6") convolution GEMMs: fp32-accurate products on the bf16 matrix cores.  Forward / data gradient with tap
// reuse (clconv_x6t_kernel), weight gradient with transposing LDS reads (clconv_wgrad_x6_kernel), the weight pre-split, and
// their C entry points.  The fp32-MFMA counterparts and the fwd / dgrad entry points that dispatch here: ign_clconv_f32.hip.
#include "ign_clconv.h"

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
constexpr int X6_BLOCK = 3 * TN * KC;        // bf16 per packed weight block (one step of one n-tile): 3 planes x 128 rows x 16 k

// The same split on pairs: f32x2 -> bf16x2 is ONE v_cvt_pk_bf16_f32 and the halves come out packed, so a pair costs
// 3 conversions + 4 unpacks + 4 subtractions (5.5 VALU per element; the scalar form above compiles to ~10).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(f32x2 v, bf16x2& x0, bf16x2& x1, bf16x2& x2) {
    x0 = __builtin_convertvector(v, bf16x2);
    f32x2 r = v - __builtin_convertvector(x0, f32x2);
    x1 = __builtin_convertvector(r, bf16x2);
    r -= __builtin_convertvector(x1, f32x2);
    x2 = __builtin_convertvector(r, bf16x2);
}
// split V consecutive values and write them to three LDS planes `plane` elements apart
template <int V>
__device__ __forceinline__ void split3_store(const float (&t)[V], __bf16* d, int plane);
template <>
__device__ __forceinline__ void split3_store<1>(const float (&t)[1], __bf16* d, int plane);
__device__ __forceinline__ void split3(float v, __bf16& x0, __bf16& x1, __bf16& x2) {
    x0 = (__bf16)v;
    float r = v - (float)x0;
    x1 = (__bf16)r;
    r -= (float)x1;
    x2 = (__bf16)r;
}
template <>
__device__ __forceinline__ void split3_store<1>(const float (&t)[1], __bf16* d, int plane) {
    __bf16 x0, x1, x2;
    split3(t[0], x0, x1, x2);
    d[0] = x0; d[plane] = x1; d[2 * plane] = x2;
}
template <>
__device__ __forceinline__ void split3_store<2>(const float (&t)[2], __bf16* d, int plane) {
    bf16x2 x0, x1, x2;
    split3_pair(f32x2{t[0], t[1]}, x0, x1, x2);
    *reinterpret_cast<bf16x2*>(d) = x0;
    *reinterpret_cast<bf16x2*>(d + plane) = x1;
    *reinterpret_cast<bf16x2*>(d + 2 * plane) = x2;
}
template <>
__device__ __forceinline__ void split3_store<4>(const float (&t)[4], __bf16* d, int plane) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x2 a0, a1, a2, b0, b1, b2;
    split3_pair(f32x2{t[0], t[1]}, a0, a1, a2);
    split3_pair(f32x2{t[2], t[3]}, b0, b1, b2);
    *reinterpret_cast<bf16x4*>(d) = __builtin_shufflevector(a0, b0, 0, 1, 2, 3);
    *reinterpret_cast<bf16x4*>(d + plane) = __builtin_shufflevector(a1, b1, 0, 1, 2, 3);
    *reinterpret_cast<bf16x4*>(d + 2 * plane) = __builtin_shufflevector(a2, b2, 0, 1, 2, 3);
}

// Two fp16 planes: x = h0 + h1 + O(2^-22 |x|), h0 = fp16(x), h1 = fp16(x - h0) (the residual is an exact fp32 subtraction).
// The caller has scaled x by a power of two so that h0 stays below fp16's overflow and h1 stays a normal number for typical
// elements (ign_pow2_scale).  Planes are 16-bit slots of the same LDS layout as the bf16 planes.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h_pair(f32x2 v, f16x2& x0, f16x2& x1) {
    x0 = __builtin_convertvector(v, f16x2);
    x1 = __builtin_convertvector(v - __builtin_convertvector(x0, f32x2), f16x2);
}
template <int V>
__device__ __forceinline__ void split2h_store(const float (&t)[V], __bf16* d, int plane) {
    _Float16* dh = reinterpret_cast<_Float16*>(d);
    if constexpr (V == 1) {
        const _Float16 x0 = (_Float16)t[0];
        dh[0] = x0; dh[plane] = (_Float16)(t[0] - (float)x0);
    } else if constexpr (V == 2) {
        f16x2 x0, x1;
        split2h_pair(f32x2{t[0], t[1]}, x0, x1);
        *reinterpret_cast<f16x2*>(dh) = x0;
        *reinterpret_cast<f16x2*>(dh + plane) = x1;
    } else {
        f16x2 a0, a1, b0, b1;
        split2h_pair(f32x2{t[0], t[1]}, a0, a1);
        split2h_pair(f32x2{t[2], t[3]}, b0, b1);
        *reinterpret_cast<f16x4*>(dh) = __builtin_shufflevector(a0, b0, 0, 1, 2, 3);
        *reinterpret_cast<f16x4*>(dh + plane) = __builtin_shufflevector(a1, b1, 0, 1, 2, 3);
    }
}

// NP = number of 16-bit planes an operand is split into: 3 (bf16, six partial products, fp32 accuracy), 2 (fp16 planes of a
// power-of-two-scaled operand, three partial products, fp32 accuracy) or 1 (the operand rounded to bf16, one product: the
// arithmetic of the reference's default bf16-autocast mode, fp32 accumulation).
template <int V, int NP>
__device__ __forceinline__ void split_store(const float (&t)[V], __bf16* d, int plane) {
    if constexpr (NP == 3) {
        split3_store<V>(t, d, plane);
    } else if constexpr (NP == 2) {
        split2h_store<V>(t, d, plane);
    } else if constexpr (V == 1) {
        d[0] = (__bf16)t[0];
    } else if constexpr (V == 2) {
        *reinterpret_cast<bf16x2*>(d) = __builtin_convertvector(f32x2{t[0], t[1]}, bf16x2);
    } else {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        const bf16x2 a = __builtin_convertvector(f32x2{t[0], t[1]}, bf16x2), b = __builtin_convertvector(f32x2{t[2], t[3]}, bf16x2);
        *reinterpret_cast<bf16x4*>(d) = __builtin_shufflevector(a, b, 0, 1, 2, 3);
    }
}

// one 32x32x16 MFMA step on a pair of 16-bit operand fragments: bf16 planes (NP 3 / 1) or fp16 planes (NP 2)
template <int NP>
__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (NP == 2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ---- split-bf16 convolution with tap reuse.  The im2col rows of 128 consecutive output positions of one sample overlap:
// together they are the (128 + k - 1)-row span of the input.  Per 16-channel chunk that span is loaded, passed through the
// prologue and split ONCE, and all k taps run from it with the MFMA row operand shifted by one LDS row per tap -- the plain
// GEMM form above loads and splits every activation k times.  Weights: bf16 planes (3, N, k*Cp), Cp = channels rounded up to
// 16, so every (tap, chunk) block is 16-byte aligned.  One barrier per (chunk, tap) step of 24 MFMAs per wave; the weights
// of the next step and (on the last tap) the next span are prefetched into registers during the MFMAs.
#ifndef IGN_H3_WAVES
#define IGN_H3_WAVES 3                             // minimum waves per SIMD asked of the fp16-plane convolution kernel (A/B: -DIGN_H3_WAVES=2)
#endif
constexpr int X6T_SPAN = TM + 15;                  // k <= 16
constexpr int X6T_APLANE = X6T_SPAN * X6_PITCH;

// LDS of one workgroup: two stages of NPL operand planes each (NPL = 2 for the fp16 arithmetic: 52 KB, three workgroups per CU
// -- the register budget below is set to match; 3 otherwise: 78 KB, two workgroups per CU)
template <int NP> struct X6tLds {
    static constexpr int NPL = (NP == 2) ? 2 : 3;
    static constexpr int ABUF = NPL * X6T_APLANE, BBUF = NPL * X6_PLANE;
    static constexpr size_t BYTES = (size_t)2 * (ABUF + BBUF) * sizeof(unsigned short);
};
// Three workgroups per CU for the fp16 forward kernel (168 VGPRs, 3 spilled: 0.733 -> 0.678 ms for the FCN's forward convolutions on
// one box); the data-gradient variant keeps two -- its epilogue holds 16 y values per accumulator and spills 99 registers at the
// 168 budget (0.505 -> 0.667 ms).
template <int NP, int EPI> struct X6tWaves { static constexpr int N = (NP == 2 && EPI == EPI_BIAS_STATS) ? IGN_H3_WAVES : 2; };
template <int V, bool PRO, int EPI, int NP>
__global__ void __launch_bounds__(256, (X6tWaves<NP, EPI>::N)) clconv_x6t_kernel(const ConvX6Args ca) {
    const GemmNTArgs& a = ca.g;
    constexpr int VPR = KC / V;                               // vectors per span row
    constexpr int APASS = (X6T_SPAN * VPR + 255) / 256;       // staging passes of the span (V=4: 3, V=2: 5, V=1: 9)
    constexpr int X6T_ABUF = X6tLds<NP>::ABUF, X6T_BBUF = X6tLds<NP>::BBUF;      // (shadow the three-plane constants)
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
    // weights are packed step-block-major (ign_clconv_pack_weights_x3): the 3 planes x 128 rows x 16 k of one (n-tile, chunk,
    // tap) step are ONE contiguous 12 KB block, so a wave's 16-byte loads cover 1 KB of consecutive addresses (the row-major
    // layout made every 32-byte row piece pull its own 128-byte line from L2: 4x the traffic, and the issue of the three
    // loads alone took ~1000 cycles per step)
    const unsigned short* bsrc = a.B3 + (size_t)nt * (size_t)(ca.cp / KC) * ca.k * X6_BLOCK + (size_t)brw * KC + 8 * bh;
    constexpr size_t bplane = (size_t)TN * KC;

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
    uint4 rb00, rb01 = {}, rb02 = {}, rb10, rb11 = {}, rb12 = {};
    const int ncc = ca.cp / KC;
    const int nstep = ncc * ca.k;

    // two-plane fp16 path: operand scales (powers of two) from device-side bounds; the epilogue divides them out again
    const float sa = (NP == 2) ? ign_pow2_scale(ca.bound_a) : 1.f;
    const float osc = (NP == 2) ? 1.f / (sa * ign_pow2_scale(ca.bound_b)) : 1.f;
    auto aload = [&](int cc) {
        const int ch = cc * KC + (tid % VPR) * V;          // 256 % VPR == 0: a thread keeps its channel offset in every pass
        a_ok = ch < ca.cin;                                  // V divides cin: a vector is entirely inside or outside
        if (PRO && a_ok) {
            vload<V>(pa, a.pro_a + ch); vload<V>(pb, a.pro_b + ch);
            if constexpr (NP == 2) {                         // relu(sa (a y + b)) = sa relu(a y + b): the scale rides on a, b
#pragma unroll
                for (int v = 0; v < V; ++v) { pa[v] *= sa; pb[v] *= sa; }
            }
        }
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
                float tv[V];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    tv[v] = ra[p][v];
                    if (PRO && a_ok) tv[v] = fmaxf(fmaf(pa[v], tv[v], pb[v]), 0.f);
                    else if (NP == 2) tv[v] *= sa;
                }
                split_store<V, NP>(tv, st + row * X6_PITCH + q * V, X6T_APLANE);
            }
        }
    };
    // step index -> offset of its block; steps past the end re-read the last block (the loads stay unconditional so that
    // the compiler's vmcnt bookkeeping is exact on every path)
    auto boff = [&](int st) {
        st = min(st, nstep - 1);
        return (size_t)st * X6_BLOCK;            // step = chunk * k + tap
    };
    auto bstore = [&](int buf, const uint4& r0, const uint4& r1, const uint4& r2) {
        __bf16* st = Bbuf + buf * X6T_BBUF + brw * X6_PITCH + 8 * bh;
        *reinterpret_cast<uint4*>(st) = r0;
        if constexpr (NP >= 2) *reinterpret_cast<uint4*>(st + X6_PLANE) = r1;
        if constexpr (NP == 3) *reinterpret_cast<uint4*>(st + 2 * X6_PLANE) = r2;
    };
#define IGN_BLOAD(r0, r1, r2, st)                                                                       \
    do {                                                                                                \
        const size_t off_ = boff(st);                                                                   \
        r0 = *reinterpret_cast<const uint4*>(bsrc + off_);                                              \
        if constexpr (NP >= 2) r1 = *reinterpret_cast<const uint4*>(bsrc + bplane + off_);              \
        if constexpr (NP == 3) r2 = *reinterpret_cast<const uint4*>(bsrc + 2 * bplane + off_);          \
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
        bf16x8 af[2][NP], bf[2][NP];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                af[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * X6T_APLANE + i * 32 * X6_PITCH);
                bf[i][pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * X6_PLANE + i * 32 * X6_PITCH);
            }
        // keep the twelve fragment reads together in front of the MFMAs (one exposed LDS latency per step instead of the
        // three the scheduler otherwise creates by interleaving read groups with the first MFMAs), and the LDS stores of
        // the prefetched operands behind them
        __builtin_amdgcn_sched_barrier(0);
#define IGN_X6(pa_, pb_)                                                                                   \
        acc[0][0] = mfma16<NP>(af[0][pa_], bf[0][pb_], acc[0][0]);                                         \
        acc[0][1] = mfma16<NP>(af[0][pa_], bf[1][pb_], acc[0][1]);                                         \
        acc[1][0] = mfma16<NP>(af[1][pa_], bf[0][pb_], acc[1][0]);                                         \
        acc[1][1] = mfma16<NP>(af[1][pa_], bf[1][pb_], acc[1][1]);
        if constexpr (NP == 3) { IGN_X6(2, 0) IGN_X6(0, 2) IGN_X6(1, 1) IGN_X6(1, 0) IGN_X6(0, 1) }
        if constexpr (NP == 2) { IGN_X6(1, 0) IGN_X6(0, 1) }
        IGN_X6(0, 0)
#undef IGN_X6
        __builtin_amdgcn_sched_barrier(0);
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
    nt_epilogue<EPI, NP == 2>(a, acc, reinterpret_cast<float*>(smem16), mt, m0, n0, bi * ca.trows + min(ca.trows, t0 + TM), osc);
}

// ---- the k = 1 case with wide outputs (a Linear layer: N % 256 == 0): 128 x 256 output tile, eight waves (2 x 4 of 64 x 64).
// A Linear layer has no taps to share a staged span, so in clconv_x6t_kernel every 24-MFMA step pays for loading, splitting and
// storing a 128 x 16 activation tile; on this part those instructions do not hide under other waves' MFMAs, they add
// to them.  Sharing the activation tile between twice as many waves halves that cost per MFMA: per wave and step one float4 to
// load / split / store instead of two, the weight side unchanged.  Same packed weights (two consecutive 128-row n-tiles), same
// arithmetic and summation order per output, same epilogue code.
// LDS sized by the planes actually staged (NP = 2: two fp16 planes = 74 KB, so two workgroups fit a CU's 160 KB; three bf16 planes
// = 111 KB, one workgroup); IGN_H3W_WAVES = waves per SIMD the register budget of the two-plane variant is pinned to.
#ifndef IGN_H3W_WAVES
#define IGN_H3W_WAVES 2
#endif
template <int NP> struct X6wLds {
    static constexpr int NPL = NP == 3 ? 3 : (NP == 2 ? 2 : 1);
    static constexpr int ABUF = NPL * TM * X6_PITCH;             // one 128-row activation tile
    static constexpr int BBUF = 2 * NPL * X6_PLANE;              // two 128-row weight blocks
    static constexpr size_t BYTES = (size_t)2 * (ABUF + BBUF) * sizeof(unsigned short);
    static constexpr int WAVES = NP == 2 ? IGN_H3W_WAVES : 2;
};

template <int NP, int EPI = EPI_BIAS_STATS>
__global__ void __launch_bounds__(512, X6wLds<NP>::WAVES) clconv_x6w_kernel(const ConvX6Args ca) {
    const GemmNTArgs& a = ca.g;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    constexpr int X6W_ABUF = X6wLds<NP>::ABUF, X6W_BBUF = X6wLds<NP>::BBUF, NPL = X6wLds<NP>::NPL;
    __bf16* Abuf = reinterpret_cast<__bf16*>(smem16);
    __bf16* Bbuf = Abuf + 2 * X6W_ABUF;
    constexpr int APLANE = TM * X6_PITCH;

    const int ntw = a.N / 256;                                    // wide n-tiles
    const int nwg = a.mtiles * ntw;
    int lid = blockIdx.x;
    {
        const int per = nwg / 8;
        if (lid < per * 8) lid = (lid & 7) * per + (lid >> 3);
    }
    const int mt = lid / ntw, nw = lid - mt * ntw;
    const int m0 = mt * TM, n0 = nw * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;                      // wn 0..3
    const int ncc = ca.cp / KC;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // activation staging: thread -> float4 (row tid >> 2, channels 4 * (tid & 3)); two register sets by step parity, loaded two
    // steps ahead like the weights
    const int arow = tid >> 2, ach = (tid & 3) * 4;
    const float* ap = a.A + (long long)min(m0 + arow, a.M - 1) * ca.cin + ach;
    float ra0[4], ra1[4];
    // weight staging: threads 0..255 -> n-tile 2*nw, 256..511 -> n-tile 2*nw + 1; row (tid & 255) >> 1, 8-channel half tid & 1
    const int bt = tid >> 8, brw = (tid & 255) >> 1, bh = tid & 1;
    const unsigned short* bsrc = a.B3 + (size_t)(2 * nw + bt) * (size_t)ncc * X6_BLOCK + (size_t)brw * KC + 8 * bh;
    constexpr size_t bplane = (size_t)TN * KC;
    uint4 rb00, rb01 = {}, rb02 = {}, rb10, rb11 = {}, rb12 = {};

    const float sa = (NP == 2) ? ign_pow2_scale(ca.bound_a) : 1.f;
    const float osc = (NP == 2) ? 1.f / (sa * ign_pow2_scale(ca.bound_b)) : 1.f;
#define IGN_ALOADW(ra_, cc_)                                                                 \
    do {                                                                                     \
        const int c_ = min((cc_), ncc - 1);                                                  \
        if (c_ * KC + ach < ca.cin) vload<4>(ra_, ap + c_ * KC);  /* cin % 4 == 0 */         \
        else ra_[0] = ra_[1] = ra_[2] = ra_[3] = 0.f;                                        \
    } while (0)
    auto astore = [&](int buf, const float (&r)[4]) {
        if constexpr (NP == 2) {
            const float rs[4] = {r[0] * sa, r[1] * sa, r[2] * sa, r[3] * sa};
            split_store<4, NP>(rs, Abuf + buf * X6W_ABUF + arow * X6_PITCH + ach, APLANE);
        } else {
            split_store<4, NP>(r, Abuf + buf * X6W_ABUF + arow * X6_PITCH + ach, APLANE);
        }
    };
    auto bstore = [&](int buf, const uint4& r0, const uint4& r1, const uint4& r2) {
        __bf16* st = Bbuf + buf * X6W_BBUF + bt * NPL * X6_PLANE + brw * X6_PITCH + 8 * bh;
        *reinterpret_cast<uint4*>(st) = r0;
        if constexpr (NP >= 2) *reinterpret_cast<uint4*>(st + X6_PLANE) = r1;
        if constexpr (NP == 3) *reinterpret_cast<uint4*>(st + 2 * X6_PLANE) = r2;
    };
#define IGN_BLOADW(r0, r1, r2, st)                                                                      \
    do {                                                                                                \
        const size_t off_ = (size_t)min((st), ncc - 1) * X6_BLOCK;                                      \
        r0 = *reinterpret_cast<const uint4*>(bsrc + off_);                                              \
        if constexpr (NP >= 2) r1 = *reinterpret_cast<const uint4*>(bsrc + bplane + off_);              \
        if constexpr (NP == 3) r2 = *reinterpret_cast<const uint4*>(bsrc + 2 * bplane + off_);          \
    } while (0)

    IGN_ALOADW(ra0, 0);
    IGN_BLOADW(rb00, rb01, rb02, 0);
    IGN_ALOADW(ra1, 1);
    IGN_BLOADW(rb10, rb11, rb12, 1);
    astore(0, ra0);
    bstore(0, rb00, rb01, rb02);
    __syncthreads();

    // Waves w and w + 4 share a SIMD and run the same program between the same barriers.  Waves 4..7 do the step's stores BEFORE
    // its MFMAs, waves 0..3 after (MI355X_MICROARCH.md, "try a stagger").  Measured neutral here (135/158/182/176 vs
    // 136/157/183/177 TFLOP/s-equivalent): SQ_VALU_MFMA_COEXEC_CYCLES stays at 1.4 % of SQ_VALU_MFMA_BUSY_CYCLES either way
    // (profiles/r1m_pmc_gemm_coexec.json) -- the idle matrix-pipe time is barrier / s_waitcnt wait (32 % of wave cycles).
    const bool late = wave >= 4;
    auto mm = [&](int step) {
        const __bf16* As = Abuf + (step & 1) * X6W_ABUF + (wm * 64 + l31) * X6_PITCH + 8 * h;
        const __bf16* Bs = Bbuf + (step & 1) * X6W_BBUF + (wn >> 1) * NPL * X6_PLANE + ((wn & 1) * 64 + l31) * X6_PITCH + 8 * h;
        bf16x8 af[2][NP], bf[2][NP];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                af[i][pl] = *reinterpret_cast<const bf16x8*>(As + pl * APLANE + i * 32 * X6_PITCH);
                bf[i][pl] = *reinterpret_cast<const bf16x8*>(Bs + pl * X6_PLANE + i * 32 * X6_PITCH);
            }
        __builtin_amdgcn_sched_barrier(0);
#define IGN_X6(pa_, pb_)                                                                                   \
        acc[0][0] = mfma16<NP>(af[0][pa_], bf[0][pb_], acc[0][0]);                                         \
        acc[0][1] = mfma16<NP>(af[0][pa_], bf[1][pb_], acc[0][1]);                                         \
        acc[1][0] = mfma16<NP>(af[1][pa_], bf[0][pb_], acc[1][0]);                                         \
        acc[1][1] = mfma16<NP>(af[1][pa_], bf[1][pb_], acc[1][1]);
        if constexpr (NP == 3) { IGN_X6(2, 0) IGN_X6(0, 2) IGN_X6(1, 1) }
        if constexpr (NP >= 2) { IGN_X6(1, 0) IGN_X6(0, 1) }
        IGN_X6(0, 0)
#undef IGN_X6
        __builtin_amdgcn_sched_barrier(0);
    };
    auto body = [&](int step, bool odd) {
        // the register sets of this step's parity hold data that is already in LDS: refill them for step + 2
        if (odd) { IGN_BLOADW(rb10, rb11, rb12, step + 2); IGN_ALOADW(ra1, step + 2); }
        else { IGN_BLOADW(rb00, rb01, rb02, step + 2); IGN_ALOADW(ra0, step + 2); }
        const bool next = step + 1 < ncc;
        auto st = [&]() {
            if (odd) { bstore((step + 1) & 1, rb00, rb01, rb02); if (next) astore((step + 1) & 1, ra0); }
            else { bstore((step + 1) & 1, rb10, rb11, rb12); if (next) astore((step + 1) & 1, ra1); }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (late) { st(); mm(step); } else { mm(step); st(); }
        __syncthreads();
    };
    for (int step = 0; step < ncc; step += 2) {
        body(step, false);
        if (step + 1 < ncc) body(step + 1, true);
    }
#undef IGN_BLOADW
#undef IGN_ALOADW
    // epilogue: bias + store (nt_epilogue_body derives wm / wn from the thread index: wn 0..3 covers the 256 columns)
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    const int m_lim = a.M;
    if (m0 + TM <= m_lim) nt_epilogue_body<EPI, true, NP == 2>(a, acc, s1, s2, m0, n0, m_lim, osc);
    else nt_epilogue_body<EPI, false, NP == 2>(a, acc, s1, s2, m0, n0, m_lim, osc);
}

// Step-block-major planes for the kernel above: block (n-tile, chunk cc, tap j) = [plane][row 0..127][16 channels of chunk cc]
// of tap j, contiguous; rows past N and channels past C are zero.
//   forward: rows = co, value W[co][ci][j];   data gradient: rows = ci, channels = co, value W[co][ci][k-1-jj].
// `bound` (nullable): pack for the two-plane fp16 kernels instead -- planes 0 / 1 hold the fp16 split of w * ign_pow2_scale(bound)
// (plane 2 of the block is left untouched; the layout is the three-plane one so that both paths share their addressing).
__device__ __forceinline__ void pack_weights_x3t_body(const float* __restrict__ w, unsigned short* __restrict__ wt3,
                                                      unsigned short* __restrict__ wd3, int Co, int Ci, int k, int Cip, int Cop,
                                                      long long i, const float* __restrict__ bound = nullptr) {
    const long long nf = (long long)((Co + TN - 1) / TN) * (Cip / KC) * k * (TN * KC);        // elements per plane set / 3
    const long long nd = wd3 ? (long long)((Ci + TN - 1) / TN) * (Cop / KC) * k * (TN * KC) : 0;
    const bool fwd = i < nf;
    if (!fwd && i >= nf + nd) return;
    const long long e = fwd ? i : i - nf;
    // e = ((nt * ncc + cc) * k + j) * (128*16) + row * 16 + q
    const int q = (int)(e % KC);
    long long t = e / KC;
    const int row = (int)(t % TN);
    t /= TN;
    const int j = (int)(t % k);
    t /= k;
    const int ncc = (fwd ? Cip : Cop) / KC;
    const int cc = (int)(t % ncc);
    const int nt = (int)(t / ncc);
    const int n = nt * TN + row, c = cc * KC + q;
    float v = 0.f;
    if (fwd) { if (n < Co && c < Ci) v = w[((long long)n * Ci + c) * k + j]; }
    else     { if (n < Ci && c < Co) v = w[((long long)c * Ci + n) * k + (k - 1 - j)]; }
    const long long blk = ((long long)(nt * ncc + cc) * k + j) * X6_BLOCK + (long long)row * KC + q;
    if (bound) {
        const float sv = v * ign_pow2_scale(bound);
        const _Float16 h0 = (_Float16)sv;
        _Float16* dh = reinterpret_cast<_Float16*>(fwd ? wt3 : wd3) + blk;
        dh[0] = h0; dh[TN * KC] = (_Float16)(sv - (float)h0);
        return;
    }
    __bf16 x0, x1, x2;
    split3(v, x0, x1, x2);
    __bf16* dst = reinterpret_cast<__bf16*>(fwd ? wt3 : wd3) + blk;
    dst[0] = x0; dst[TN * KC] = x1; dst[2 * TN * KC] = x2;
}

__global__ void __launch_bounds__(256) pack_weights_x3t_kernel(const float* __restrict__ w, unsigned short* __restrict__ wt3,
                                                               unsigned short* __restrict__ wd3, int Co, int Ci, int k, int Cip,
                                                               int Cop) {
    pack_weights_x3t_body(w, wt3, wd3, Co, Ci, k, Cip, Cop, (long long)blockIdx.x * 256 + threadIdx.x);
}

// The weights of every layer of a convolution stack in ONE launch (blockIdx.y = layer), plus the per-step counter bump of the
// BatchNorm behind each layer (nn.BatchNorm1d.num_batches_tracked += 1 in training mode): the "prologue" of an FCN step.
constexpr int PACK_LMAX = 8;
struct PackMultiTable {
    const float* w[PACK_LMAX];
    unsigned short* wt3[PACK_LMAX];
    unsigned short* wd3[PACK_LMAX];
    long long* counter[PACK_LMAX];
    const float* bound[PACK_LMAX];             // non-null: two-plane fp16 packing, scaled by ign_pow2_scale(bound)
    int Co[PACK_LMAX], Ci[PACK_LMAX], k[PACK_LMAX];
};
__global__ void __launch_bounds__(256) pack_weights_x3t_multi_kernel(const PackMultiTable t) {
    const int l = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0 && t.counter[l]) *t.counter[l] += 1;
    pack_weights_x3t_body(t.w[l], t.wt3[l], t.wd3[l], t.Co[l], t.Ci[l], t.k[l], (t.Ci[l] + 15) / 16 * 16, (t.Co[l] + 15) / 16 * 16,
                          (long long)blockIdx.x * 256 + threadIdx.x, t.bound[l]);
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
    const float* bound_dy; const float* bound_x;                   // NP = 2 (fp16 planes): operand bounds (ign_pow2_scale), else null
};

__device__ __forceinline__ bf16x8 lds_tr8(const __bf16* p0, const __bf16* p1) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int KT, int NR, int VX, bool PRO, int NP>
// (fp16 arithmetic: two planes leave room for a third workgroup per CU; the 8-tap variant keeps two -- eight accumulators do not
//  fit the 168-register budget of three waves per SIMD without ~100 spilled registers)
__global__ void __launch_bounds__(256, ((NP == 2 && KT <= 5) ? IGN_H3_WAVES : 2)) clconv_wgrad_x6_kernel(const WgradX6Args a) {
    constexpr int RU = 16 * NR;                     // output rows per unit
    constexpr int SPAN = RU + KT - 1;               // input rows per unit
    constexpr int PPLANE = RU * WG_PITCH, QPLANE = SPAN * WG_PITCH;
    constexpr int NPL = (NP == 2) ? 2 : 3;          // planes held per operand (two for the fp16 arithmetic: 2/3 of the LDS)
    constexpr int STAGE = NPL * (PPLANE + QPLANE);
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
    // two-plane fp16 path: both operands are scaled by powers of two before the split, the partial sums are scaled back
    const float sdy = (NP == 2) ? ign_pow2_scale(a.bound_dy) : 1.f;
    const float sx = (NP == 2) ? ign_pow2_scale(a.bound_x) : 1.f;
    const float osc = (NP == 2) ? 1.f / (sdy * sx) : 1.f;
    if (NP == 2 && PRO && x_ok) {
#pragma unroll
        for (int v = 0; v < VX; ++v) { pa[v] *= sx; pb[v] *= sx; }
    }

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
        __bf16* Q = P + NPL * PPLANE;
#pragma unroll
        for (int p = 0; p < DPASS; ++p) {
            if constexpr (NP == 2) {
#pragma unroll
                for (int v = 0; v < 4; ++v) rd[p][v] *= sdy;
            }
            split_store<4, NP>(rd[p], P + (dr + 16 * p) * WG_PITCH + dc, PPLANE);
        }
#pragma unroll
        for (int p = 0; p < XPASS; ++p) {
            const int row = (tid + p * 256) / XVPR;
            if (row < SPAN) {
                float tv[VX];
#pragma unroll
                for (int v = 0; v < VX; ++v) {
                    tv[v] = rx[p][v];
                    if (PRO && x_ok) tv[v] = fmaxf(fmaf(pa[v], tv[v], pb[v]), 0.f);
                    else if (NP == 2) tv[v] *= sx;
                }
                split_store<VX, NP>(tv, Q + row * WG_PITCH + xq, QPLANE);
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
        const __bf16* Q = P + NPL * PPLANE;
#pragma unroll
        for (int s = 0; s < NR; ++s) {
            bf16x8 af[NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const __bf16* base = P + pl * PPLANE + (16 * s + row_lo) * WG_PITCH + acol;
                af[pl] = lds_tr8(base, base + 4 * WG_PITCH);
            }
            // software pipeline over the taps: the transposed reads of tap j+1 are in flight while tap j's six MFMAs run
            bf16x8 bf[2][NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const __bf16* base = Q + pl * QPLANE + (16 * s + row_lo) * WG_PITCH + bcol;
                bf[0][pl] = lds_tr8(base, base + 4 * WG_PITCH);
            }
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                if (j + 1 < KT) {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        const __bf16* base = Q + pl * QPLANE + (16 * s + row_lo + j + 1) * WG_PITCH + bcol;
                        bf[(j + 1) & 1][pl] = lds_tr8(base, base + 4 * WG_PITCH);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8(&b)[NP] = bf[j & 1];
                if constexpr (NP == 3) {
                    acc[j] = mfma16<NP>(af[2], b[0], acc[j]);
                    acc[j] = mfma16<NP>(af[0], b[2], acc[j]);
                    acc[j] = mfma16<NP>(af[1], b[1], acc[j]);
                }
                if constexpr (NP >= 2) {
                    acc[j] = mfma16<NP>(af[1], b[0], acc[j]);
                    acc[j] = mfma16<NP>(af[0], b[1], acc[j]);
                }
                acc[j] = mfma16<NP>(af[0], b[0], acc[j]);
                __builtin_amdgcn_sched_barrier(0);
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
                if (co < a.Co) out[((long long)co * a.k + j) * a.Ci + ci] = (NP == 2) ? acc[j][r] * osc : acc[j][r];
            }
    }
}

// ---- the k = 1 case (a Linear layer: dW[co][ci] = sum_m dy[m][co] * x[m][ci]) has no taps to share a staged span, so the
// tile grows instead: 128 co x 128 ci per workgroup, wave 64 x 64 = four accumulators, units of 16 rows, 24 MFMAs per wave and
// barrier.  Same staging (row-major, one split per element), same transposing reads, same fixed-order reduction of the splits.
constexpr int W1_PITCH = 128 + 32;            // bf16 per staged row: 320 B; rows 0..3 of a block start 16 banks apart
constexpr int W1_PLANE = 16 * W1_PITCH;

struct Wgrad1Args {
    const float* dy; const float* x; float* part;      // dy (M, Co), x (M, Ci), part (nsplit, Co, Ci)
    float* part_b;                                     // (nsplit, Co) column sums of dy (the bias gradient) or null
    long long M;
    int Ci, Co, nunits, nsplit, citiles, ntiles;
    const float* bound_dy; const float* bound_x;        // NP = 2 (fp16 planes): operand bounds (ign_pow2_scale), else null
};

template <int NP>
__global__ void __launch_bounds__(256, (NP == 2 ? IGN_H3_WAVES : 2)) clconv_wgrad_x6_k1_kernel(const Wgrad1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    __bf16* smem = reinterpret_cast<__bf16*>(smem16);
    constexpr int NPL = (NP == 2) ? 2 : 3;                        // planes per operand held in LDS
    constexpr int W1_STAGE = 2 * NPL * W1_PLANE;                  // (shadows the three-plane constant)
    // Workgroups go round-robin over the 8 XCDs: XCD x takes the row ranges (splits) = x mod 8 and runs all tiles of a split
    // back to back, so the dy / x rows a split shares between its tiles are fetched into ONE L2, once.
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int tile = seq % a.ntiles;
    const int split = (seq / a.ntiles) * 8 + xcd;
    if (split >= a.nsplit) return;
    const int cot = tile / a.citiles, cit = tile - cot * a.citiles;
    const int co0 = cot * 128, ci0 = cit * 128;
    const int per = (a.nunits + a.nsplit - 1) / a.nsplit;
    const int u_begin = split * per, u_end = min(a.nunits, u_begin + per);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wco = wave & 1, wci = wave >> 1;

    const int sr = tid >> 5, sc = (tid & 31) * 4;             // staging: rows sr and sr + 8, four channels from sc
    const bool d_ok = co0 + sc < a.Co, x_ok = ci0 + sc < a.Ci;        // Co % 4 == 0, Ci % 4 == 0
    const float* dyp = a.dy + co0 + sc;
    const float* xp = a.x + ci0 + sc;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float rd[2][4], rx[2][4];
    const float sdy = (NP == 2) ? ign_pow2_scale(a.bound_dy) : 1.f;
    const float sx = (NP == 2) ? ign_pow2_scale(a.bound_x) : 1.f;
    const float osc = (NP == 2) ? 1.f / (sdy * sx) : 1.f;
    // bias gradient = column sums of dy: the tiles of the first ci-column see every dy value of their co-block once
    const bool want_b = a.part_b != nullptr && cit == 0;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int u) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const long long m = (long long)u * 16 + sr + 8 * p;
            if (d_ok && m < a.M) vload<4>(rd[p], dyp + m * a.Co);
            else { rd[p][0] = rd[p][1] = rd[p][2] = rd[p][3] = 0.f; }
            if (x_ok && m < a.M) vload<4>(rx[p], xp + m * a.Ci);
            else { rx[p][0] = rx[p][1] = rx[p][2] = rx[p][3] = 0.f; }
        }
    };
    auto lstore = [&](int buf) {
        __bf16* P = smem + buf * W1_STAGE;
        __bf16* Q = P + NPL * W1_PLANE;
        if (want_b) {
#pragma unroll
            for (int v = 0; v < 4; ++v) bsum[v] += rd[0][v] + rd[1][v];
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if constexpr (NP == 2) {
#pragma unroll
                for (int v = 0; v < 4; ++v) { rd[p][v] *= sdy; rx[p][v] *= sx; }
            }
            split_store<4, NP>(rd[p], P + (sr + 8 * p) * W1_PITCH + sc, W1_PLANE);
            split_store<4, NP>(rx[p], Q + (sr + 8 * p) * W1_PITCH + sc, W1_PLANE);
        }
    };

    // transposed-read coordinates as in clconv_wgrad_x6_kernel
    const int u16 = lane & 15, q4 = u16 >> 2, p4 = u16 & 3, G1 = (lane >> 4) & 1;
    const int roff = (8 * h + q4) * W1_PITCH + 16 * G1 + 4 * p4;

    if (u_begin < u_end) {
        gload(u_begin);
        lstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int u = u_begin; u < u_end; ++u, buf ^= 1) {
        if (u + 1 < u_end) gload(u + 1);
        const __bf16* P = smem + buf * W1_STAGE + roff + wco * 64;
        const __bf16* Q = smem + buf * W1_STAGE + NPL * W1_PLANE + roff + wci * 64;
        bf16x8 bf[2][NP], af[2][NP];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                bf[j][pl] = lds_tr8(Q + pl * W1_PLANE + 32 * j, Q + pl * W1_PLANE + 32 * j + 4 * W1_PITCH);
                af[j][pl] = lds_tr8(P + pl * W1_PLANE + 32 * j, P + pl * W1_PLANE + 32 * j + 4 * W1_PITCH);
            }
        __builtin_amdgcn_sched_barrier(0);
#define IGN_W1(pa_, pb_)                                                                                   \
        acc[0][0] = mfma16<NP>(af[0][pa_], bf[0][pb_], acc[0][0]);                                         \
        acc[0][1] = mfma16<NP>(af[0][pa_], bf[1][pb_], acc[0][1]);                                         \
        acc[1][0] = mfma16<NP>(af[1][pa_], bf[0][pb_], acc[1][0]);                                         \
        acc[1][1] = mfma16<NP>(af[1][pa_], bf[1][pb_], acc[1][1]);
        if constexpr (NP == 3) { IGN_W1(2, 0) IGN_W1(0, 2) IGN_W1(1, 1) }
        if constexpr (NP >= 2) { IGN_W1(1, 0) IGN_W1(0, 1) }
        IGN_W1(0, 0)
#undef IGN_W1
        __builtin_amdgcn_sched_barrier(0);
        if (u + 1 < u_end) lstore(buf ^ 1);
        __syncthreads();
    }

    if (want_b) {
        // the eight threads that staged rows sr = 0..7 of the same four columns: fixed-order sum through LDS
        float* red = reinterpret_cast<float*>(smem16);           // every wave is past its last operand read (loop barrier)
        *reinterpret_cast<float4*>(red + sr * 128 + sc) = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
        __syncthreads();
        if (tid < 128 && co0 + tid < a.Co) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += red[r * 128 + tid];
            a.part_b[(long long)split * a.Co + co0 + tid] = t;          // (accumulated from the unscaled dy values)
        }
    }
    float* out = a.part + (long long)split * a.Co * a.Ci;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = ci0 + wci * 64 + 32 * j + l31;
        if (ci < a.Ci) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wco * 64 + 32 * i + acc_row16(r, h);
                    if (co < a.Co) out[(long long)co * a.Ci + ci] = (NP == 2) ? acc[i][j][r] * osc : acc[i][j][r];
                }
        }
    }
}

// dW[co][ci][j] = sum_s part[s][co][j*Ci + ci]   (s ascending: bitwise reproducible); torch (Co, Ci, k) layout
// ------------------------------------------------------------------------------------------------ C ABI
template <int EPI, int NP>
static int launch_x6t(const ConvX6Args& a, int V, bool pro, hipStream_t s) {
    const dim3 grid((unsigned)(a.g.mtiles * a.g.ntiles)), block(256);
#define IGN_X6T(VV, PP)                                                                                                      \
    do {                                                                                                                     \
        static bool once = false;                                                                                            \
        if (!once) {                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&clconv_x6t_kernel<VV, PP, EPI, NP>),                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)X6tLds<NP>::BYTES);                   \
            once = true;                                                                                                     \
        }                                                                                                                    \
        hipLaunchKernelGGL((clconv_x6t_kernel<VV, PP, EPI, NP>), grid, block, X6tLds<NP>::BYTES, s, a);                       \
    } while (0)
    if (V == 4) { if (pro) IGN_X6T(4, true); else IGN_X6T(4, false); }
    else if (V == 2) { if (pro) IGN_X6T(2, true); else IGN_X6T(2, false); }
    else { if (pro) IGN_X6T(1, true); else IGN_X6T(1, false); }
#undef IGN_X6T
    return ign_check_launch("clconv_x6t_kernel");
}

template <int NP, int EPI = EPI_BIAS_STATS>
static int launch_x6w(const ConvX6Args& a, hipStream_t s) {
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&clconv_x6w_kernel<NP, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)X6wLds<NP>::BYTES);
        once = true;
    }
    const unsigned nwg = (unsigned)(a.g.mtiles * (a.g.N / 256));
    hipLaunchKernelGGL((clconv_x6w_kernel<NP, EPI>), dim3(nwg), dim3(512), X6wLds<NP>::BYTES, s, a);
    return ign_check_launch("clconv_x6w_kernel");
}

int ign_clconv_launch_x6t(const ConvX6Args& a, int epi, int V, bool pro, hipStream_t s) {
    if (epi == EPI_GELU_BWD) {
        // dense layer's input gradient with the GELU derivative in the epilogue: the wide-tile kernel on fp16 planes only
        if (!(a.k == 1 && a.g.N % 256 == 0 && V == 4 && !pro && !a.g.part && a.tps * 1 == a.g.mtiles && a.trows == a.g.M && a.nprod == 3
              && a.g.ey)) {
            ign_set_error("ign_clconv_launch_x6t: the GELU-gradient epilogue needs k = 1, N %% 256 == 0, C %% 4 == 0, the f16x3 arithmetic");
            return IGN_E_UNSUP;
        }
        return launch_x6w<2, EPI_GELU_BWD>(a, s);
    }
    // Linear layers with wide outputs: the 128 x 256 tile kernel (see clconv_x6w_kernel)
    if (a.k == 1 && a.g.N % 256 == 0 && V == 4 && !pro && epi == EPI_BIAS_STATS && !a.g.part &&
        a.tps * 1 == a.g.mtiles && a.trows == a.g.M)
        return a.nprod == 1 ? launch_x6w<1>(a, s) : a.nprod == 3 ? launch_x6w<2>(a, s) : launch_x6w<3>(a, s);
    if (a.nprod == 1)
        return epi == EPI_BIAS_STATS ? launch_x6t<EPI_BIAS_STATS, 1>(a, V, pro, s) : launch_x6t<EPI_MASK_STATS, 1>(a, V, pro, s);
    if (a.nprod == 3)
        return epi == EPI_BIAS_STATS ? launch_x6t<EPI_BIAS_STATS, 2>(a, V, pro, s) : launch_x6t<EPI_MASK_STATS, 2>(a, V, pro, s);
    return epi == EPI_BIAS_STATS ? launch_x6t<EPI_BIAS_STATS, 3>(a, V, pro, s) : launch_x6t<EPI_MASK_STATS, 3>(a, V, pro, s);
}

extern "C" int ign_clconv_kpad(int C) { return (C + 15) / 16 * 16; }
// bf16 elements of a packed weight set with `rows` GEMM rows (output channels) and `chans` reduction channels
extern "C" long long ign_clconv_x3_elems(int rows, int chans, int k) {
    return 3LL * ((rows + TN - 1) / TN) * TN * k * ((chans + 15) / 16 * 16);
}
extern "C" long long ign_clconv_x6_mtiles(int B, int rows) { return (long long)B * ((rows + TM - 1) / TM); }

extern "C" int ign_clconv_pack_weights_x3(const float* w_oik, void* wt3_fwd, void* wt3_dgrad, int Co, int Ci, int k, void* stream) {
    if (!w_oik || !wt3_fwd || Co <= 0 || Ci <= 0 || k <= 0) {
        ign_set_error("ign_clconv_pack_weights_x3: bad argument (Co=%d Ci=%d k=%d)", Co, Ci, k);
        return IGN_E_ARG;
    }
    const int Cip = ign_clconv_kpad(Ci), Cop = ign_clconv_kpad(Co);
    const long long n = ign_clconv_x3_elems(Co, Ci, k) / 3 + (wt3_dgrad ? ign_clconv_x3_elems(Ci, Co, k) / 3 : 0);
    hipLaunchKernelGGL(pack_weights_x3t_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oik,
                       (unsigned short*)wt3_fwd, (unsigned short*)wt3_dgrad, Co, Ci, k, Cip, Cop);
    return ign_check_launch("pack_weights_x3t_kernel");
}

static int pack_multi_impl(int n, const float* const* w_oik, void* const* wt3_fwd, void* const* wt3_dgrad, const int* Co, const int* Ci,
                           const int* k, long long* const* counters, const float* const* bounds, void* stream);

extern "C" int ign_clconv_pack_weights_x3_multi(int n, const float* const* w_oik, void* const* wt3_fwd, void* const* wt3_dgrad,
                                               const int* Co, const int* Ci, const int* k, long long* const* counters, void* stream) {
    return pack_multi_impl(n, w_oik, wt3_fwd, wt3_dgrad, Co, Ci, k, counters, nullptr, stream);
}

extern "C" int ign_clconv_pack_weights_h2_multi(int n, const float* const* w_oik, void* const* wt_fwd, void* const* wt_dgrad,
                                               const int* Co, const int* Ci, const int* k, long long* const* counters,
                                               const float* const* w_bounds, void* stream) {
    if (!w_bounds) { ign_set_error("ign_clconv_pack_weights_h2_multi: null bound table"); return IGN_E_ARG; }
    for (int l = 0; l < n && l < PACK_LMAX; ++l)
        if (!w_bounds[l]) { ign_set_error("ign_clconv_pack_weights_h2_multi: layer %d: null weight bound", l); return IGN_E_ARG; }
    return pack_multi_impl(n, w_oik, wt_fwd, wt_dgrad, Co, Ci, k, counters, w_bounds, stream);
}

static int pack_multi_impl(int n, const float* const* w_oik, void* const* wt3_fwd, void* const* wt3_dgrad, const int* Co, const int* Ci,
                           const int* k, long long* const* counters, const float* const* bounds, void* stream) {
    if (n <= 0 || n > PACK_LMAX || !w_oik || !wt3_fwd || !Co || !Ci || !k) {
        ign_set_error("ign_clconv_pack_weights_x3_multi: n=%d outside 1..%d or null table", n, PACK_LMAX);
        return IGN_E_ARG;
    }
    PackMultiTable t;
    long long nmax = 0;
    for (int l = 0; l < n; ++l) {
        if (!w_oik[l] || !wt3_fwd[l] || Co[l] <= 0 || Ci[l] <= 0 || k[l] <= 0) {
            ign_set_error("ign_clconv_pack_weights_x3_multi: layer %d: bad argument (Co=%d Ci=%d k=%d)", l, Co[l], Ci[l], k[l]);
            return IGN_E_ARG;
        }
        t.w[l] = w_oik[l]; t.wt3[l] = (unsigned short*)wt3_fwd[l]; t.wd3[l] = wt3_dgrad ? (unsigned short*)wt3_dgrad[l] : nullptr;
        t.counter[l] = counters ? counters[l] : nullptr;
        t.bound[l] = bounds ? bounds[l] : nullptr;
        t.Co[l] = Co[l]; t.Ci[l] = Ci[l]; t.k[l] = k[l];
        const long long e = ign_clconv_x3_elems(Co[l], Ci[l], k[l]) / 3 + (t.wd3[l] ? ign_clconv_x3_elems(Ci[l], Co[l], k[l]) / 3 : 0);
        nmax = e > nmax ? e : nmax;
    }
    hipLaunchKernelGGL(pack_weights_x3t_multi_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, t);
    return ign_check_launch("pack_weights_x3t_multi_kernel");
}

template <int KT, int NR, int NP>
static int launch_wgrad_x6(const WgradX6Args& a, int V, bool pro, dim3 grid, hipStream_t s) {
    constexpr size_t lds = (size_t)2 * (NP == 2 ? 2 : 3) * ((16 * NR) + (16 * NR + KT - 1)) * WG_PITCH * sizeof(unsigned short);
#define IGN_WG(VV, PP)                                                                                                       \
    do {                                                                                                                     \
        static bool once = false;                                                                                            \
        if (!once) {                                                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&clconv_wgrad_x6_kernel<KT, NR, VV, PP, NP>),             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
            once = true;                                                                                                     \
        }                                                                                                                    \
        hipLaunchKernelGGL((clconv_wgrad_x6_kernel<KT, NR, VV, PP, NP>), grid, dim3(256), lds, s, a);                         \
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

static int wgrad_x6_k1_splits(long long M, int Ci, int Co, int* nunits, int* tiles) {
    *nunits = (int)((M + 15) / 16);
    *tiles = ((Co + 127) / 128) * ((Ci + 127) / 128);
    int s = (512 + *tiles - 1) / *tiles;                      // ~2 workgroups per CU
    const int max_s = (*nunits + 31) / 32;                    // at least 32 units per split
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}

extern "C" size_t ign_clconv_wgrad_x6_workspace_bytes(int B, int Tin, int Ci, int Co, int k) {
    if (k == 1 && B > 0 && Tin > 0 && Ci > 0 && Co > 0) {
        int nunits, tiles;
        return (size_t)wgrad_x6_k1_splits((long long)B * Tin, Ci, Co, &nunits, &tiles) * Co * (Ci + 1) * sizeof(float);  // + bias partials
    }
    const int Tout = Tin - k + 1, ru = wgrad_x6_rows_per_unit(k);
    if (B <= 0 || Tout <= 0 || Ci <= 0 || Co <= 0 || !ru) return 0;
    const int tiles = ((Co + 63) / 64) * ((Ci + 63) / 64);
    const int nunits = B * ((Tout + ru - 1) / ru);
    return (size_t)wgrad_x6_splits(nunits, tiles) * Co * k * Ci * sizeof(float);
}

template <int NP>
static int wgrad_x6_impl(const char* who, const float* dyp, int dy_pad, const float* x, const float* pro_a, const float* pro_b,
                         float* dw_oik, void* workspace, int B, int Tin, int Ci, int Co, int k, void* stream,
                         float* db = nullptr, const float* bound_dy = nullptr, const float* bound_x = nullptr) {
    const int Tout = Tin - k + 1, ru = wgrad_x6_rows_per_unit(k);
    const bool defer = dw_oik == nullptr && k > 1;       // partials only: the caller reduces several layers in one launch
    if (!dyp || !x || (!dw_oik && !defer) || !workspace || B <= 0 || Ci <= 0 || Co <= 0 || k <= 0 || Tout <= 0 || dy_pad < 0 ||
        ((pro_a == nullptr) != (pro_b == nullptr))) {
        ign_set_error("%s: bad argument (B=%d Tin=%d Ci=%d Co=%d k=%d pad=%d)", who, B, Tin, Ci, Co, k, dy_pad);
        return IGN_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (k == 1) {
        // a Linear layer: rows of all samples are one (B*Tin, C) matrix (no padding rows, no prologue)
        if (Co % 4 || Ci % 4 || dy_pad || pro_a) {
            ign_set_error("%s: k = 1 needs Co %% 4 == 0, Ci %% 4 == 0, no padding, no prologue (Co=%d Ci=%d)", who, Co, Ci);
            return IGN_E_UNSUP;
        }
        Wgrad1Args w{};
        w.dy = dyp; w.x = x; w.part = (float*)workspace; w.M = (long long)B * Tin; w.Ci = Ci; w.Co = Co;
        int tiles;
        w.nsplit = wgrad_x6_k1_splits(w.M, Ci, Co, &w.nunits, &tiles);
        w.citiles = (Ci + 127) / 128; w.ntiles = tiles;
        w.part_b = db ? (float*)workspace + (size_t)w.nsplit * Co * Ci : nullptr;
        w.bound_dy = bound_dy; w.bound_x = bound_x;
        constexpr size_t w1_lds = (size_t)2 * 2 * (NP == 2 ? 2 : 3) * W1_PLANE * sizeof(unsigned short);     // two stages x two operands
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&clconv_wgrad_x6_k1_kernel<NP>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)w1_lds);
            once = true;
        }
        {
            IgnScopedTimer tm("clconv_wgrad", s);
            hipLaunchKernelGGL(clconv_wgrad_x6_k1_kernel<NP>, dim3((unsigned)(tiles * ((w.nsplit + 7) / 8 * 8))), dim3(256), w1_lds,
                               s, w);
        }
        int rc1 = ign_check_launch("clconv_wgrad_x6_k1_kernel");
        if (rc1) return rc1;
        if (db && (rc1 = ign_clconv_launch_wgrad_reduce(w.part_b, db, w.nsplit, Co, 1, 1, s))) return rc1;
        return ign_clconv_launch_wgrad_reduce((const float*)workspace, dw_oik, w.nsplit, Co, Ci, 1, s);
    }
    if (Co % 4 || !ru) { ign_set_error("%s: needs Co %% 4 == 0 and k in {1,2,3,5,8} (Co=%d k=%d)", who, Co, k); return IGN_E_UNSUP; }
    WgradX6Args a{};
    a.dy = dyp; a.dy_sample_pitch = (long long)(Tout + 2 * dy_pad) * Co; a.dy_row0 = dy_pad;
    a.x = x; a.x_sample_pitch = (long long)Tin * Ci; a.pro_a = pro_a; a.pro_b = pro_b;
    a.part = (float*)workspace; a.B = B; a.Tin = Tin; a.Tout = Tout; a.Ci = Ci; a.Co = Co; a.k = k;
    a.cps = (Tout + ru - 1) / ru; a.nunits = B * a.cps;
    a.bound_dy = bound_dy; a.bound_x = bound_x;
    a.citiles = (Ci + 63) / 64;
    const int tiles = ((Co + 63) / 64) * a.citiles;
    a.nsplit = wgrad_x6_splits(a.nunits, tiles);
    const dim3 grid((unsigned)tiles, (unsigned)a.nsplit);
    const int V = ign_vec_width(Ci);
    const bool pro = pro_a != nullptr;
    int rc;
    {
        IgnScopedTimer tm("clconv_wgrad", s);
        if (k == 8) rc = launch_wgrad_x6<8, 1, NP>(a, V, pro, grid, s);
        else if (k == 5) rc = launch_wgrad_x6<5, 1, NP>(a, V, pro, grid, s);
        else if (k == 3) rc = launch_wgrad_x6<3, 2, NP>(a, V, pro, grid, s);
        else rc = launch_wgrad_x6<2, 2, NP>(a, V, pro, grid, s);
    }
    if (rc) return rc;
    if (defer) return 0;
    return ign_clconv_launch_wgrad_reduce((const float*)workspace, dw_oik, a.nsplit, Co, Ci, k, s);
}

extern "C" int ign_clconv_wgrad_x6_nsplit(int B, int Tin, int Ci, int Co, int k) {
    const int Tout = Tin - k + 1, ru = wgrad_x6_rows_per_unit(k);
    if (B <= 0 || Tout <= 0 || Ci <= 0 || Co <= 0 || !ru) return 0;
    return wgrad_x6_splits(B * ((Tout + ru - 1) / ru), ((Co + 63) / 64) * ((Ci + 63) / 64));
}


extern "C" int ign_clconv_wgrad_x6(const float* dyp, int dy_pad, const float* x, const float* pro_a, const float* pro_b,
                                   float* dw_oik, void* workspace, int B, int Tin, int Ci, int Co, int k, void* stream) {
    return wgrad_x6_impl<3>("ign_clconv_wgrad_x6", dyp, dy_pad, x, pro_a, pro_b, dw_oik, workspace, B, Tin, Ci, Co, k, stream);
}

// The same kernel with ONE product per MFMA step (operands rounded to bf16, fp32 accumulation): the arithmetic of the
// reference's default bf16-autocast mode.  Workspace as for ign_clconv_wgrad_x6.
extern "C" int ign_clconv_wgrad_bf16(const float* dyp, int dy_pad, const float* x, const float* pro_a, const float* pro_b,
                                     float* dw_oik, void* workspace, int B, int Tin, int Ci, int Co, int k, void* stream) {
    return wgrad_x6_impl<1>("ign_clconv_wgrad_bf16", dyp, dy_pad, x, pro_a, pro_b, dw_oik, workspace, B, Tin, Ci, Co, k, stream);
}

// Two fp16 planes, three products (see include/ign_abi.h, "h3"): bound_dy / bound_x are the device-side magnitude bounds of the
// two operands (after the prologue for x).  Workspace and result as for ign_clconv_wgrad_x6.
extern "C" int ign_clconv_wgrad_h3(const float* dyp, int dy_pad, const float* x, const float* pro_a, const float* pro_b,
                                   float* dw_oik, void* workspace, const float* bound_dy, const float* bound_x, int B, int Tin,
                                   int Ci, int Co, int k, void* stream) {
    if (!bound_dy || !bound_x) { ign_set_error("ign_clconv_wgrad_h3: null operand bound"); return IGN_E_ARG; }
    return wgrad_x6_impl<2>("ign_clconv_wgrad_h3", dyp, dy_pad, x, pro_a, pro_b, dw_oik, workspace, B, Tin, Ci, Co, k, stream, nullptr,
                            bound_dy, bound_x);
}
extern "C" int ign_linear_wgrad_h3(const float* dy, const float* x, float* dw, float* db, void* workspace, const float* bound_dy,
                                   const float* bound_x, long long M, int Ci, int Co, void* stream) {
    if (M <= 0 || M > 0x7fffffffLL) { ign_set_error("ign_linear_wgrad_h3: M = %lld rows out of range", M); return IGN_E_ARG; }
    if (!bound_dy || !bound_x) { ign_set_error("ign_linear_wgrad_h3: null operand bound"); return IGN_E_ARG; }
    return wgrad_x6_impl<2>("ign_linear_wgrad_h3", dy, 0, x, nullptr, nullptr, dw, workspace, 1, (int)M, Ci, Co, 1, stream, db, bound_dy,
                            bound_x);
}

// A Linear layer's weight AND bias gradient in one pass over dy: dW = dy^T x, db = column sums of dy (the tiles of the first
// ci-column accumulate them from the dy values they stage anyway).  dy (M, Co), x (M, Ci); workspace as for
// ign_clconv_wgrad_x6 with (B, Tin, k) = (1, M, 1).
extern "C" int ign_linear_wgrad_x6(const float* dy, const float* x, float* dw, float* db, void* workspace, long long M, int Ci,
                                   int Co, void* stream) {
    if (M <= 0 || M > 0x7fffffffLL) { ign_set_error("ign_linear_wgrad_x6: M = %lld rows out of range", M); return IGN_E_ARG; }
    return wgrad_x6_impl<3>("ign_linear_wgrad_x6", dy, 0, x, nullptr, nullptr, dw, workspace, 1, (int)M, Ci, Co, 1, stream, db);
}
extern "C" int ign_linear_wgrad_bf16(const float* dy, const float* x, float* dw, float* db, void* workspace, long long M, int Ci,
                                     int Co, void* stream) {
    if (M <= 0 || M > 0x7fffffffLL) { ign_set_error("ign_linear_wgrad_bf16: M = %lld rows out of range", M); return IGN_E_ARG; }
    return wgrad_x6_impl<1>("ign_linear_wgrad_bf16", dy, 0, x, nullptr, nullptr, dw, workspace, 1, (int)M, Ci, Co, 1, stream, db);
}

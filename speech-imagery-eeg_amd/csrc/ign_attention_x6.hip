// Attention core  softmax(scale * Q K^T) V  on the bf16 matrix cores at fp32 accuracy (forward).
//
// Same maths and interface as attn_fwd_kernel (ign_attention.hip; IGN/layers/SelfAttention_Family.py:56-75), but both
// products run as split-bf16 GEMMs (csrc/ign_clconv_x6.hip): every fp32 operand is split exactly into three bf16 terms and the
// six partial products of weight >= 2^-16 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- 24 MFMAs of 32 cycles per
// 32x32 score tile and product instead of 32 fp32 MFMAs of 64 cycles, error at the fp32 kernel's level.
//   * K and V tiles are split while they are staged (fp32 global -> registers -> three bf16 LDS planes): the L/128 workgroups
//     of a (batch, head) repeat that split, but it is 5.5 VALU per element against 96 MFMAs per tile and wave, it reads 4 bytes
//     per element instead of the 6 of pre-split planes, and needs no workspace or pre-pass (a pre-split pass cost 0.7 of 4.2 ms).
//     The next tile's loads are issued before the current tile's MFMAs and land in registers behind them.  Q rows are split in
//     registers by the wave that owns them, scale folded in first.
//   * S^T = K Q^T as before (keys in the accumulator registers, the query on the lane): the online softmax is unchanged.
//   * P^T is split in registers.  Accumulator register r of lane half h holds key (r&3) + 8(r>>2) + 4h, so registers 8s..8s+7
//     are the 16-key slab s with k-slot i <-> key 16s + 4h + (i&3) + 8(i>>2): exactly two groups of 4 CONSECUTIVE keys, which is
//     what gfx950's transposing LDS read ds_read_b64_tr_b16 delivers for the A operand V^T (lane = feature column, 4 rows per
//     read) from a V tile staged row-major.  No LDS round trip for P, no transposed copy of V.
#include "ign_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct AttnX6Args {
    const float *q, *k, *v;                  // (B,L,H,E) / (B,S,H,E) with element strides sb (batch), sl (sequence); head stride E
    float *out, *lse_out;
    long long q_sb, q_sl, k_sb, k_sl, v_sb, v_sl;
    int B, L, S, H, E;
    float scale;
    const float *bq, *bk, *bv;               // NP = 2 (two fp16 planes, three products): device-side bounds of |q|, |k|, |v|
};

// ---- NP = 2: two fp16 planes of power-of-two-scaled operands, THREE products (include/ign_abi.h, "h3").  Scales come from
// device-side magnitude bounds (pow2_scale); scores are un-scaled inside the exp2 (an FMA instead of a subtraction), the
// probabilities (<= 1) are split after a multiplication by 2^14, products of bounds give hard bounds for derived operands
// (|dS| <= 2 E max|dO| max|V|).  The planes are 16-bit slots of the same LDS / register layouts as the bf16 planes.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NP>
__device__ __forceinline__ f32x16 mfma16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    if constexpr (NP == 2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#define MFMA16(a, b, c) mfma16<NP>((a), (b), (c))
__device__ __forceinline__ float pow2_scale_v(float b) {      // 2^e with 2^13 <= b 2^e < 2^14 (1 for zero / non-finite b)
    if (!(b > 0.f) || !(b < INFINITY)) return 1.f;
    int e;
    (void)frexpf(b, &e);
    e = 14 - e;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    return ldexpf(1.f, e);
}
__device__ __forceinline__ void split2h_pair(f32x2 v, f16x2& x0, f16x2& x1) {
    x0 = __builtin_convertvector(v, f16x2);
    x1 = __builtin_convertvector(v - __builtin_convertvector(x0, f32x2), f16x2);
}

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// x = x0 + x1 + x2 (bf16, round to nearest; the residuals are exact fp32 subtractions) for a pair of values
__device__ __forceinline__ void split3_pair(f32x2 v, bf16x2& x0, bf16x2& x1, bf16x2& x2) {
    x0 = __builtin_convertvector(v, bf16x2);
    f32x2 r = v - __builtin_convertvector(x0, f32x2);
    x1 = __builtin_convertvector(r, bf16x2);
    r -= __builtin_convertvector(x1, f32x2);
    x2 = __builtin_convertvector(r, bf16x2);
}
// eight values -> one bf16x8 MFMA operand per plane.  NP = 3: the exact three-way split (six products, fp32 accuracy);
// NP = 1: the values rounded to bf16 (one product: the arithmetic of the reference's bf16-autocast mode); p1, p2 stay unused.
template <int NP>
__device__ __forceinline__ void splitN_x8(const float (&t)[8], bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    if constexpr (NP == 2) {
        f16x2 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) split2h_pair(f32x2{t[2 * i], t[2 * i + 1]}, a[i], b[i]);
        p0 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(__builtin_shufflevector(a[0], a[1], 0, 1, 2, 3),
                                                                __builtin_shufflevector(a[2], a[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7));
        p1 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(__builtin_shufflevector(b[0], b[1], 0, 1, 2, 3),
                                                                __builtin_shufflevector(b[2], b[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7));
        p2 = p0;
        return;
    }
    if constexpr (NP == 1) {
        bf16x2 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = __builtin_convertvector(f32x2{t[2 * i], t[2 * i + 1]}, bf16x2);
        p0 = __builtin_shufflevector(__builtin_shufflevector(a[0], a[1], 0, 1, 2, 3), __builtin_shufflevector(a[2], a[3], 0, 1, 2, 3),
                                     0, 1, 2, 3, 4, 5, 6, 7);
        p1 = p0; p2 = p0;
        return;
    }
    bf16x2 a[4], b[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split3_pair(f32x2{t[2 * i], t[2 * i + 1]}, a[i], b[i], c[i]);
    p0 = __builtin_shufflevector(__builtin_shufflevector(a[0], a[1], 0, 1, 2, 3), __builtin_shufflevector(a[2], a[3], 0, 1, 2, 3),
                                 0, 1, 2, 3, 4, 5, 6, 7);
    p1 = __builtin_shufflevector(__builtin_shufflevector(b[0], b[1], 0, 1, 2, 3), __builtin_shufflevector(b[2], b[3], 0, 1, 2, 3),
                                 0, 1, 2, 3, 4, 5, 6, 7);
    p2 = __builtin_shufflevector(__builtin_shufflevector(c[0], c[1], 0, 1, 2, 3), __builtin_shufflevector(c[2], c[3], 0, 1, 2, 3),
                                 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ bf16x8 lds_tr8(const __bf16* p0, const __bf16* p1) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// four values -> NP LDS planes `plane` elements apart (8-byte stores)
template <int NP>
__device__ __forceinline__ void splitN_store4(const float4 t, __bf16* d, int plane) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    if constexpr (NP == 2) {
        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
        f16x2 a0, a1, b0, b1;
        split2h_pair(f32x2{t.x, t.y}, a0, a1);
        split2h_pair(f32x2{t.z, t.w}, b0, b1);
        _Float16* dh = reinterpret_cast<_Float16*>(d);
        *reinterpret_cast<f16x4*>(dh) = __builtin_shufflevector(a0, b0, 0, 1, 2, 3);
        *reinterpret_cast<f16x4*>(dh + plane) = __builtin_shufflevector(a1, b1, 0, 1, 2, 3);
        return;
    }
    if constexpr (NP == 1) {
        const bf16x2 a = __builtin_convertvector(f32x2{t.x, t.y}, bf16x2), b = __builtin_convertvector(f32x2{t.z, t.w}, bf16x2);
        *reinterpret_cast<bf16x4*>(d) = __builtin_shufflevector(a, b, 0, 1, 2, 3);
        return;
    }
    bf16x2 a0, a1, a2, b0, b1, b2;
    split3_pair(f32x2{t.x, t.y}, a0, a1, a2);
    split3_pair(f32x2{t.z, t.w}, b0, b1, b2);
    *reinterpret_cast<bf16x4*>(d) = __builtin_shufflevector(a0, b0, 0, 1, 2, 3);
    *reinterpret_cast<bf16x4*>(d + plane) = __builtin_shufflevector(a1, b1, 0, 1, 2, 3);
    *reinterpret_cast<bf16x4*>(d + 2 * plane) = __builtin_shufflevector(a2, b2, 0, 1, 2, 3);
}

template <int E> struct AxPitch {
    static constexpr int KT = E == 16 ? 64 : 32;                  // keys staged per LDS tile (256 threads cover >= 1 pass)
    static constexpr int K = E + 8;                               // 16-byte reads of 8 consecutive e: (E+8)*2 B = 16 * odd
    static constexpr int V = E >= 64 ? E + 32 : E + E / 2;        // transposing reads: rows 0..3 of a block 16 banks apart
};

template <int E, int NP>
__global__ void __launch_bounds__(256, 2) attn_fwd_x6_kernel(const AttnX6Args a) {
    constexpr int NS = E / 16, ED = (E + 31) / 32, PK = AxPitch<E>::K, PV = AxPitch<E>::V, AX_KT = AxPitch<E>::KT;
    constexpr int KPLANE = AX_KT * PK, VPLANE = AX_KT * PV;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    __bf16* Ks = reinterpret_cast<__bf16*>(smem16);
    __bf16* Vs = Ks + 3 * KPLANE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int qi = blockIdx.x * 128 + wave * 32 + l31;
    const bool q_ok = qi < a.L;

    // Q^T operand: lane (query, h) holds e = 16s + 8h .. +7 of slab s, three planes, scale folded in before the split
    const float sc2u = a.scale * 1.44269504088896341f;
    // NP = 2: operand scales; `us` un-scales a score inside the exp2, `pexp` = 14 makes the exp2 return 2^14 p
    const float sq = (NP == 2) ? pow2_scale_v(*a.bq * fabsf(sc2u)) : 1.f;
    const float sk = (NP == 2) ? pow2_scale_v(*a.bk) : 1.f;
    const float sv = (NP == 2) ? pow2_scale_v(*a.bv) : 1.f;
    const float us = (NP == 2) ? 1.f / (sq * sk) : 1.f;
    constexpr float pexp = (NP == 2) ? 14.f : 0.f;
    const float sc2 = sc2u * sq;
    bf16x8 Qf[3][NS];
    {
        const float* qp = a.q + b * a.q_sb + (long long)(q_ok ? qi : a.L - 1) * a.q_sl + head * E + 8 * h;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 t0 = *reinterpret_cast<const float4*>(qp + 16 * s);
            const float4 t1 = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
            // scale * log2(e) folded in: the softmax below runs in base 2 (v_exp_f32 is 2^x: no multiply per score)
            const float t[8] = {t0.x * sc2, t0.y * sc2, t0.z * sc2, t0.w * sc2, t1.x * sc2, t1.y * sc2, t1.z * sc2, t1.w * sc2};
            splitN_x8<NP>(t, Qf[0][s], Qf[1][s], Qf[2][s]);
        }
    }
    f32x16 O[ED];
#pragma unroll
    for (int d = 0; d < ED; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    // transposed-read coordinates of the V^T operand (see the header): lane supplies row q4 / columns 4*p4 of its 16-lane
    // group's block and receives column 16*G1 + u16 = l31
    const int u16 = lane & 15, q4 = u16 >> 2, p4 = u16 & 3, G1 = (lane >> 4) & 1;
    const int voff = (4 * h + q4) * PV + 16 * G1 + 4 * p4;
    const float* kbase = a.k + b * a.k_sb + head * E;
    const float* vbase = a.v + b * a.v_sb + head * E;
    // staging: thread -> float4 pieces (row sr + 16*p, columns sc..sc+3), p < NP4; 256 threads cover 16 rows x E/4... per pass
    constexpr int V4 = E / 4, RPP = 256 / V4, NP4 = AX_KT / RPP;      // rows per pass, passes per tile
    const int sr = threadIdx.x / V4, sc = (threadIdx.x - sr * V4) * 4;
    float4 rk[NP4], rv[NP4];
#define IGN_GLOAD(kt_)                                                                               \
    _Pragma("unroll") for (int p = 0; p < NP4; ++p) {                                                \
        const int r_ = (kt_) + sr + RPP * p;                                                         \
        const int rc_ = min(r_, a.S - 1);                                                            \
        float4 tk_ = *reinterpret_cast<const float4*>(kbase + (long long)rc_ * a.k_sl + sc);         \
        float4 tv_ = *reinterpret_cast<const float4*>(vbase + (long long)rc_ * a.v_sl + sc);         \
        if (r_ >= a.S) tk_ = tv_ = make_float4(0.f, 0.f, 0.f, 0.f);                                  \
        if constexpr (NP == 2) {                                                                     \
            tk_ = make_float4(tk_.x * sk, tk_.y * sk, tk_.z * sk, tk_.w * sk);                       \
            tv_ = make_float4(tv_.x * sv, tv_.y * sv, tv_.z * sv, tv_.w * sv);                       \
        }                                                                                            \
        rk[p] = tk_; rv[p] = tv_;                                                                    \
    }
    IGN_GLOAD(0)
    for (int kt0 = 0; kt0 < a.S; kt0 += AX_KT) {
        __syncthreads();                                           // every wave is done with the previous tile
#pragma unroll
        for (int p = 0; p < NP4; ++p) {
            splitN_store4<NP>(rk[p], Ks + (sr + RPP * p) * PK + sc, KPLANE);
            splitN_store4<NP>(rv[p], Vs + (sr + RPP * p) * PV + sc, VPLANE);
        }
        __syncthreads();
        if (kt0 + AX_KT < a.S) { IGN_GLOAD(kt0 + AX_KT) }          // lands behind this tile's MFMAs
#pragma unroll
        for (int sub = 0; sub < AX_KT / 32; ++sub) {
            const int kb = sub * 32;
            if (kt0 + kb < a.S) {
                // two partial accumulators: a chain of dependent MFMAs on ONE accumulator issues at about half rate
                f32x16 acc, acc2;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
                const __bf16* kr = Ks + (kb + l31) * PK + 8 * h;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(kr + 16 * s);
                    const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(kr + KPLANE + 16 * s);
                    const bf16x8 k2 = *reinterpret_cast<const bf16x8*>(kr + 2 * KPLANE + 16 * s);
                    if constexpr (NP == 3) {
                        acc = MFMA16(k2, Qf[0][s], acc);
                        acc2 = MFMA16(k0, Qf[2][s], acc2);
                        acc = MFMA16(k1, Qf[1][s], acc);
                        acc2 = MFMA16(k1, Qf[0][s], acc2);
                        acc = MFMA16(k0, Qf[1][s], acc);
                        acc2 = MFMA16(k0, Qf[0][s], acc2);
                    } else if constexpr (NP == 2) {
                        acc = MFMA16(k1, Qf[0][s], acc);
                        acc2 = MFMA16(k0, Qf[1][s], acc2);
                        if (s & 1) acc2 = MFMA16(k0, Qf[0][s], acc2);
                        else acc = MFMA16(k0, Qf[0][s], acc);
                    } else {
                        if (s & 1) acc2 = MFMA16(k0, Qf[0][s], acc2);
                        else acc = MFMA16(k0, Qf[0][s], acc);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
                // online softmax over this lane's 16 keys + the partner half's 16 keys
                float mloc = -INFINITY;
                if (kt0 + kb + 32 > a.S) {                       // wave-uniform: only the last tile has keys past S
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kt0 + kb + acc_row(r, h) >= a.S) acc[r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, acc[r]);
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                if constexpr (NP == 2) mloc *= us;               // the accumulators hold sq sk S: the maximum commutes with a positive scale
                const float mnew = fmaxf(m, mloc);
                const float alpha = __builtin_amdgcn_exp2f(m - mnew);
                const bool rescale = __builtin_amdgcn_ballot_w64(mnew != m) != 0;     // wave-uniform: the running max moved
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (NP == 2) acc[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], us, pexp - mnew));    // 2^14 p
                    else acc[r] = __builtin_amdgcn_exp2f(acc[r] - mnew);
                    psum += acc[r];
                }
                psum += __shfl_xor(psum, 32, 64);
                l = l * alpha + psum;
                m = mnew;
                // P^T operand: registers 8*s2 .. 8*s2+7 are the 16-key slab s2
                bf16x8 Pf[3][2];
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const float t[8] = {acc[8 * s2], acc[8 * s2 + 1], acc[8 * s2 + 2], acc[8 * s2 + 3],
                                        acc[8 * s2 + 4], acc[8 * s2 + 5], acc[8 * s2 + 6], acc[8 * s2 + 7]};
                    splitN_x8<NP>(t, Pf[0][s2], Pf[1][s2], Pf[2][s2]);
                }
                if (rescale) {                                   // after the first tiles the maximum rarely moves
#pragma unroll
                    for (int d = 0; d < ED; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 vf[ED][3];
#pragma unroll
                    for (int d = 0; d < ED; ++d) {
                        const __bf16* vp = Vs + (kb + 16 * s2) * PV + voff + d * 32;
                        vf[d][0] = lds_tr8(vp, vp + 8 * PV);
                        if constexpr (NP >= 2) vf[d][1] = lds_tr8(vp + VPLANE, vp + VPLANE + 8 * PV);
                        if constexpr (NP == 3) vf[d][2] = lds_tr8(vp + 2 * VPLANE, vp + 2 * VPLANE + 8 * PV);
                    }
                    // the feature blocks are independent accumulators: alternate them
#define IGN_PV(pa_, pb_) _Pragma("unroll") for (int d = 0; d < ED; ++d) O[d] = MFMA16(vf[d][pa_], Pf[pb_][s2], O[d]);
                    if constexpr (NP == 3) { IGN_PV(2, 0) IGN_PV(0, 2) IGN_PV(1, 1) }
                    if constexpr (NP >= 2) { IGN_PV(1, 0) IGN_PV(0, 1) }
                    IGN_PV(0, 0)
#undef IGN_PV
                }
            }
        }
    }
    if (q_ok) {
        // NP = 2: O holds sv * 2^14 * sum p v and l holds 2^14 * sum p
        const float inv = (NP == 2) ? 1.f / (l * sv) : 1.f / l;
        if constexpr (NP == 2) l *= 6.103515625e-05f;              // 2^-14: the log-sum-exp below is of the unscaled sum
        float* op = a.out + (((long long)b * a.L + qi) * a.H + head) * E;
#pragma unroll
        for (int d = 0; d < ED; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * h;
                if (d0 < E)
                    *reinterpret_cast<float4*>(op + d0) = make_float4(O[d][4 * g] * inv, O[d][4 * g + 1] * inv,
                                                                      O[d][4 * g + 2] * inv, O[d][4 * g + 3] * inv);
            }
        if (h == 0) a.lse_out[((long long)b * a.H + head) * a.L + qi] = m * 0.693147180559945309f + __logf(l);
    }
}

// ================================================================================================ backward
// Same three-kernel structure as every split-bf16 GEMM here: operands that stay fixed for a wave live in registers as bf16x8
// planes, streamed tiles are split while they are staged (32 rows x E, pitch E + 8: one layout serves both the row-wise
// 16-byte operand reads and the transposing reads), probabilities / score gradients are split in registers and are already in
// B-operand layout.  The fp32 backward keeps K^T and V^T (64 VGPRs) and both gradient accumulators in one kernel; in split
// form that is 96 + 64 + 32 + 24 + fragments > 256 VGPRs at two waves per SIMD, so dK and dV are separate kernels that each
// recompute S (MFMAs per 32x32 tile: dQ 72, dK 72, dV 48 of 32 cycles -- against 224 of 64 cycles in fp32).
struct AttnX6BwdArgs {
    const float *q, *k, *v, *o, *go, *lse, *delta; // o, go (B,L,H,E) contiguous; lse, delta (B,H,L)
    float *gq, *gk, *gv, *delta_out;               // delta_out: written by the dQ kernel (delta = rowsum(dO * O)), read by dK
    long long q_sb, q_sl, k_sb, k_sl, v_sb, v_sl;
    long long gq_sb, gkv_sb, g_sl;                 // element strides of the gradient outputs (batch for gq / gk, gv; sequence)
    int B, L, S, H, E;
    float scale;
    const float *bq, *bk, *bv, *bg;                // NP = 2: device-side bounds of |q|, |k|, |v|, |dO|
    float* gmax;                                   // nullable: max |dq|, |dk|, |dv| as an atomic maximum (bound of the packed gradient)
};

constexpr int AB_T = 32;                                          // rows per staged tile
template <int E> struct AbCfg {
    static constexpr int P = E + 8;                               // bf16 per staged row
    static constexpr int PLANE = AB_T * P;
    static constexpr int V4 = E / 4;                              // float4 pieces per row
    static constexpr int RPP = 256 / V4 < AB_T ? 256 / V4 : AB_T; // rows covered per pass by 256 threads
    static constexpr int NP4 = AB_T / RPP;
    static constexpr size_t LDS = (size_t)2 * 3 * PLANE * sizeof(unsigned short) + 2 * AB_T * sizeof(float) + 256;
};

// eight consecutive fp32 values (two float4) -> the three bf16x8 planes of an MFMA operand, optional scale
template <int NP>
__device__ __forceinline__ void load_split8(const float* p, float sc, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
    const float4 t0 = *reinterpret_cast<const float4*>(p);
    const float4 t1 = *reinterpret_cast<const float4*>(p + 4);
    const float t[8] = {t0.x * sc, t0.y * sc, t0.z * sc, t0.w * sc, t1.x * sc, t1.y * sc, t1.z * sc, t1.w * sc};
    splitN_x8<NP>(t, p0, p1, p2);
}

#define IGN_X6_PRODUCTS(acc_, a0, a1, a2, b0, b1, b2)  \
    if constexpr (NP == 3) {                           \
        acc_ = MFMA16(a2, b0, acc_);                   \
        acc_ = MFMA16(a0, b2, acc_);                   \
        acc_ = MFMA16(a1, b1, acc_);                   \
        acc_ = MFMA16(a1, b0, acc_);                   \
        acc_ = MFMA16(a0, b1, acc_);                   \
    }                                                  \
    acc_ = MFMA16(a0, b0, acc_);

// acc (rows = tile rows, column = lane) += T[32 rows][E] * F^T, T in LDS planes (row-wise reads), F this lane's register planes
template <int E, int NP>
__device__ __forceinline__ void x6_rows_times_regs(f32x16& acc, const __bf16* T, const bf16x8 (&F)[3][E / 16], int l31, int h) {
    constexpr int P = AbCfg<E>::P, PLANE = AbCfg<E>::PLANE;
    const __bf16* tr = T + l31 * P + 8 * h;
    f32x16 acc2;                      // second partial accumulator: dependent MFMAs on one accumulator issue at about half rate
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
    for (int s = 0; s < E / 16; ++s) {
        const bf16x8 t0 = *reinterpret_cast<const bf16x8*>(tr + 16 * s);
        if constexpr (NP == 3) {
            const bf16x8 t1 = *reinterpret_cast<const bf16x8*>(tr + PLANE + 16 * s);
            const bf16x8 t2 = *reinterpret_cast<const bf16x8*>(tr + 2 * PLANE + 16 * s);
            acc = MFMA16(t2, F[0][s], acc);
            acc2 = MFMA16(t0, F[2][s], acc2);
            acc = MFMA16(t1, F[1][s], acc);
            acc2 = MFMA16(t1, F[0][s], acc2);
            acc = MFMA16(t0, F[1][s], acc);
            acc2 = MFMA16(t0, F[0][s], acc2);
        } else if constexpr (NP == 2) {
            const bf16x8 t1 = *reinterpret_cast<const bf16x8*>(tr + PLANE + 16 * s);
            acc = MFMA16(t1, F[0][s], acc);
            acc2 = MFMA16(t0, F[1][s], acc2);
            if (s & 1) acc2 = MFMA16(t0, F[0][s], acc2);
            else acc = MFMA16(t0, F[0][s], acc);
        } else {
            if (s & 1) acc2 = MFMA16(t0, F[0][s], acc2);
            else acc = MFMA16(t0, F[0][s], acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
}

// G[d] (rows = feature d*32 + .., column = lane) += T^T * W, T in LDS planes (transposing reads), W = 16 register values of this
// lane over the tile's 32 rows in accumulator order (split here)
template <int E, int NP>
__device__ __forceinline__ void x6_tileT_times_acc(f32x16 (&G)[(E + 31) / 32], const __bf16* T, const f32x16& w, int lane) {
    constexpr int P = AbCfg<E>::P, PLANE = AbCfg<E>::PLANE, ED = (E + 31) / 32;
    const int h = lane >> 5, u16 = lane & 15, q4 = u16 >> 2, p4 = u16 & 3, G1 = (lane >> 4) & 1;
    const int off = (4 * h + q4) * P + 16 * G1 + 4 * p4;
    bf16x8 W[3][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const float t[8] = {w[8 * s2], w[8 * s2 + 1], w[8 * s2 + 2], w[8 * s2 + 3], w[8 * s2 + 4], w[8 * s2 + 5], w[8 * s2 + 6],
                            w[8 * s2 + 7]};
        splitN_x8<NP>(t, W[0][s2], W[1][s2], W[2][s2]);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 tf[ED][3];
#pragma unroll
        for (int d = 0; d < ED; ++d) {
            const __bf16* tp = T + (16 * s2) * P + off + d * 32;
            tf[d][0] = lds_tr8(tp, tp + 8 * P);
            if constexpr (NP >= 2) tf[d][1] = lds_tr8(tp + PLANE, tp + PLANE + 8 * P);
            if constexpr (NP == 3) tf[d][2] = lds_tr8(tp + 2 * PLANE, tp + 2 * PLANE + 8 * P);
        }
        // the feature blocks are independent accumulators: alternate them
#define IGN_TW(pa_, pb_) _Pragma("unroll") for (int d = 0; d < ED; ++d) G[d] = MFMA16(tf[d][pa_], W[pb_][s2], G[d]);
        if constexpr (NP == 3) { IGN_TW(2, 0) IGN_TW(0, 2) IGN_TW(1, 1) }
        if constexpr (NP >= 2) { IGN_TW(1, 0) IGN_TW(0, 1) }
        IGN_TW(0, 0)
#undef IGN_TW
    }
}

// rows t0 .. t0+31 of two (rows, H, E) fp32 tensors -> registers (zero past `nrows`); thread -> float4 piece (sr + RPP*p, sc)
#define IGN_AB_GLOAD(t0_, nrows_, basea_, sla_, baseb_, slb_)                                        \
    _Pragma("unroll") for (int p = 0; p < NP4; ++p) {                                                \
        const int r_ = (t0_) + sr + RPP * p;                                                         \
        const int rc_ = min(r_, (nrows_) - 1);                                                       \
        float4 ta_ = *reinterpret_cast<const float4*>((basea_) + (long long)rc_ * (sla_) + sc);      \
        float4 tb_ = *reinterpret_cast<const float4*>((baseb_) + (long long)rc_ * (slb_) + sc);      \
        if (r_ >= (nrows_) || !stg) ta_ = tb_ = make_float4(0.f, 0.f, 0.f, 0.f);                     \
        if constexpr (NP == 2) {                                                                     \
            ta_ = make_float4(ta_.x * sta, ta_.y * sta, ta_.z * sta, ta_.w * sta);                   \
            tb_ = make_float4(tb_.x * stb, tb_.y * stb, tb_.z * stb, tb_.w * stb);                   \
        }                                                                                            \
        ra[p] = ta_; rb[p] = tb_;                                                                    \
    }
#define IGN_AB_STORE(Ta_, Tb_)                                                                       \
    if (stg) {                                                                                       \
        _Pragma("unroll") for (int p = 0; p < NP4; ++p) {                                            \
            splitN_store4<NP>(ra[p], (Ta_) + (sr + RPP * p) * P + sc, PLANE);                            \
            splitN_store4<NP>(rb[p], (Tb_) + (sr + RPP * p) * P + sc, PLANE);                            \
        }                                                                                            \
    }

template <int E>
__device__ __forceinline__ float store_grad_rows(float* dst, const f32x16 (&G)[(E + 31) / 32], float sc, int h) {
    float am = 0.f;                                                  // max |value stored| by this lane
#pragma unroll
    for (int d = 0; d < (E + 31) / 32; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = d * 32 + 8 * g + 4 * h;
            if (d0 < E) {
                const float4 o = make_float4(G[d][4 * g] * sc, G[d][4 * g + 1] * sc, G[d][4 * g + 2] * sc, G[d][4 * g + 3] * sc);
                *reinterpret_cast<float4*>(dst + d0) = o;
                am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
    return am;
}
__device__ __forceinline__ void wave_amax_to(float* slot, float am, int lane) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if (lane == 0) ign_atomic_absmax(slot, am);
}

// ---- dQ: block = 4 waves x 32 queries (lanes); loops over key tiles.  S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - delta),
// dQ^T += K^T dS^T, scaled once at the end.
template <int E, int NP>
__global__ void __launch_bounds__(256, 2) attn_bwd_dq_x6_kernel(const AttnX6BwdArgs a) {
    constexpr int NS = E / 16, ED = (E + 31) / 32, P = AbCfg<E>::P, PLANE = AbCfg<E>::PLANE;
    constexpr int V4 = AbCfg<E>::V4, RPP = AbCfg<E>::RPP, NP4 = AbCfg<E>::NP4;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    __bf16* Ks = reinterpret_cast<__bf16*>(smem16);
    __bf16* Vs = Ks + 3 * PLANE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int qi = blockIdx.x * 128 + wave * 32 + l31;
    const bool q_ok = qi < a.L;
    const long long qrow = q_ok ? qi : a.L - 1;

    // NP = 2 scales: registers Q (sq, incl. the softmax scale) and dO (sg); staged tiles K (sta = sk) and V (stb = sv);
    // dS by the hard bound |dS| <= p (|dP| + |delta|) <= 2 E max|dO| max|V|
    const float sc2u = a.scale * 1.44269504088896341f;
    const float sq = (NP == 2) ? pow2_scale_v(*a.bq * fabsf(sc2u)) : 1.f;
    const float sg = (NP == 2) ? pow2_scale_v(*a.bg) : 1.f;
    const float sta = (NP == 2) ? pow2_scale_v(*a.bk) : 1.f, stb = (NP == 2) ? pow2_scale_v(*a.bv) : 1.f;
    const float sds = (NP == 2) ? pow2_scale_v(2.f * (float)E * *a.bg * *a.bv) : 1.f;
    const float us = (NP == 2) ? 1.f / (sq * sta) : 1.f, up = (NP == 2) ? 1.f / (sg * stb) : 1.f;
    bf16x8 Qf[3][NS], Gf[3][NS];
    float del_q = 0.f;                                         // delta = rowsum(dO * O): this lane's half of the row, then lane ^ 32
    {
        const float* qp = a.q + b * a.q_sb + qrow * a.q_sl + head * E + 8 * h;
        const float* gp = a.go + (((long long)b * a.L + qrow) * a.H + head) * E + 8 * h;
        const float* op = a.o + (((long long)b * a.L + qrow) * a.H + head) * E + 8 * h;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            load_split8<NP>(qp + 16 * s, sc2u * sq, Qf[0][s], Qf[1][s], Qf[2][s]);   // base-2 softmax
            const float4 g0 = *reinterpret_cast<const float4*>(gp + 16 * s), g1 = *reinterpret_cast<const float4*>(gp + 16 * s + 4);
            const float4 o0 = *reinterpret_cast<const float4*>(op + 16 * s), o1 = *reinterpret_cast<const float4*>(op + 16 * s + 4);
            del_q += g0.x * o0.x + g0.y * o0.y + g0.z * o0.z + g0.w * o0.w + g1.x * o1.x + g1.y * o1.y + g1.z * o1.z + g1.w * o1.w;
            const float t[8] = {g0.x * sg, g0.y * sg, g0.z * sg, g0.w * sg, g1.x * sg, g1.y * sg, g1.z * sg, g1.w * sg};
            splitN_x8<NP>(t, Gf[0][s], Gf[1][s], Gf[2][s]);
        }
    }
    del_q += __shfl_xor(del_q, 32, 64);
    if (q_ok && h == 0) a.delta_out[((long long)b * a.H + head) * a.L + qi] = del_q;
    const float lse_q = a.lse[((long long)b * a.H + head) * a.L + qrow] * 1.44269504088896341f;
    f32x16 dQ[ED];
#pragma unroll
    for (int d = 0; d < ED; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQ[d][r] = 0.f;

    const float* kbase = a.k + b * a.k_sb + head * E;
    const float* vbase = a.v + b * a.v_sb + head * E;
    const int sr = threadIdx.x / V4, sc = (threadIdx.x - sr * V4) * 4;
    const bool stg = sr < AB_T;
    float4 ra[NP4], rb[NP4];
    IGN_AB_GLOAD(0, a.S, kbase, a.k_sl, vbase, a.v_sl)
    for (int kt0 = 0; kt0 < a.S; kt0 += AB_T) {
        __syncthreads();
        IGN_AB_STORE(Ks, Vs)
        __syncthreads();
        if (kt0 + AB_T < a.S) { IGN_AB_GLOAD(kt0 + AB_T, a.S, kbase, a.k_sl, vbase, a.v_sl) }
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
        x6_rows_times_regs<E, NP>(st, Ks, Qf, l31, h);
        x6_rows_times_regs<E, NP>(dp, Vs, Gf, l31, h);
        // rows = keys kt0 + acc_row(r,h), column = this lane's query
        if (kt0 + AB_T > a.S) {                                  // wave-uniform: only the last tile has keys past S
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kt0 + acc_row(r, h) >= a.S) st[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (NP == 2) dp[r] = __builtin_amdgcn_exp2f(fmaf(st[r], us, -lse_q)) * fmaf(dp[r], up * sds, -del_q * sds);
            else dp[r] = __builtin_amdgcn_exp2f(st[r] - lse_q) * (dp[r] - del_q);
        }
        x6_tileT_times_acc<E, NP>(dQ, Ks, dp, lane);
    }
    // NP = 2: dQ holds sk sds K^T dS
    float am = 0.f;
    if (q_ok) am = store_grad_rows<E>(a.gq + b * a.gq_sb + (long long)qi * a.g_sl + head * E, dQ,
                                      (NP == 2) ? a.scale / (sta * sds) : a.scale, h);
    if (a.gmax) wave_amax_to(a.gmax, am, lane);
}

// ---- dK / dV: block = 4 waves x 32 keys (lanes); loops over query tiles.  S = Q K^T (rows = queries), P = exp(S - lse);
//   DV:  dV^T += dO^T P                                   (K^T in registers)
//   !DV: dP = dO V^T, dS = P (dP - delta), dK^T += Q^T dS  (K^T and V^T in registers), scaled once at the end
template <int E, bool DV, int NP>
__global__ void __launch_bounds__(256, 2) attn_bwd_dkv_x6_kernel(const AttnX6BwdArgs a) {
    constexpr int NS = E / 16, ED = (E + 31) / 32, P = AbCfg<E>::P, PLANE = AbCfg<E>::PLANE;
    constexpr int V4 = AbCfg<E>::V4, RPP = AbCfg<E>::RPP, NP4 = AbCfg<E>::NP4;
    extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
    __bf16* Qs = reinterpret_cast<__bf16*>(smem16);
    __bf16* Gs = Qs + 3 * PLANE;
    float* Ls = reinterpret_cast<float*>(Gs + 3 * PLANE);          // lse[32], delta[32] of the tile's queries
    float* Ds = Ls + AB_T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int ki = blockIdx.x * 128 + wave * 32 + l31;
    const bool k_ok = ki < a.S;
    const float kmask = k_ok ? 0.f : -INFINITY;                   // a lane past S contributes p = 2^-inf = 0
    const long long krow = k_ok ? ki : a.S - 1;

    // NP = 2 scales: registers K (sk, incl. the softmax scale) and V (sv); staged tiles Q (sta = sq) and dO (stb = sg);
    // P (<= 1) by 2^14, dS by its hard bound 2 E max|dO| max|V|
    const float sc2u = a.scale * 1.44269504088896341f;
    const float sk = (NP == 2) ? pow2_scale_v(*a.bk * fabsf(sc2u)) : 1.f;
    const float sv = (NP == 2) ? pow2_scale_v(*a.bv) : 1.f;
    const float sta = (NP == 2) ? pow2_scale_v(*a.bq) : 1.f, stb = (NP == 2) ? pow2_scale_v(*a.bg) : 1.f;
    const float sds = (NP == 2) ? pow2_scale_v(2.f * (float)E * *a.bg * *a.bv) : 1.f;
    const float us = (NP == 2) ? 1.f / (sk * sta) : 1.f, up = (NP == 2) ? 1.f / (sv * stb) : 1.f;
    const float pexp = (NP == 2 && DV) ? 14.f : 0.f;              // the dV product takes 2^14 p as its operand
    bf16x8 Kf[3][NS], Vf[3][DV ? 1 : NS];
    {
        const float* kp = a.k + b * a.k_sb + krow * a.k_sl + head * E + 8 * h;
        const float* vp = a.v + b * a.v_sb + krow * a.v_sl + head * E + 8 * h;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            load_split8<NP>(kp + 16 * s, sc2u * sk, Kf[0][s], Kf[1][s], Kf[2][s]);   // base-2 softmax
            if constexpr (!DV) load_split8<NP>(vp + 16 * s, sv, Vf[0][s], Vf[1][s], Vf[2][s]);
        }
    }
    f32x16 G[ED];
#pragma unroll
    for (int d = 0; d < ED; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) G[d][r] = 0.f;

    const float* qbase = a.q + b * a.q_sb + head * E;
    const float* gbase = a.go + (long long)b * a.L * a.H * E + head * E;
    const long long g_sl = (long long)a.H * E;
    const float* lse_b = a.lse + ((long long)b * a.H + head) * a.L;
    const float* del_b = a.delta + ((long long)b * a.H + head) * a.L;
    const int sr = threadIdx.x / V4, sc = (threadIdx.x - sr * V4) * 4;
    const bool stg = sr < AB_T;
    float4 ra[NP4], rb[NP4];
    float rl = 0.f;
    IGN_AB_GLOAD(0, a.L, qbase, a.q_sl, gbase, g_sl)
    if (threadIdx.x < 2 * AB_T) {
        const int q = threadIdx.x & (AB_T - 1);
        rl = q < a.L ? (threadIdx.x < AB_T ? lse_b[q] * 1.44269504088896341f : del_b[q]) : (threadIdx.x < AB_T ? INFINITY : 0.f);
    }
    for (int qt0 = 0; qt0 < a.L; qt0 += AB_T) {
        __syncthreads();
        IGN_AB_STORE(Qs, Gs)
        if (threadIdx.x < 2 * AB_T) Ls[threadIdx.x] = rl;            // Ls and Ds are contiguous
        __syncthreads();
        if (qt0 + AB_T < a.L) {
            IGN_AB_GLOAD(qt0 + AB_T, a.L, qbase, a.q_sl, gbase, g_sl)
            if (threadIdx.x < 2 * AB_T) {
                const int q = qt0 + AB_T + (threadIdx.x & (AB_T - 1));
                rl = q < a.L ? (threadIdx.x < AB_T ? lse_b[q] * 1.44269504088896341f : del_b[q]) : (threadIdx.x < AB_T ? INFINITY : 0.f);
            }
        }
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        x6_rows_times_regs<E, NP>(s, Qs, Kf, l31, h);
        // rows = queries qt0 + acc_row(r,h) (padded queries carry lse = inf: p = 0), column = this lane's key
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 lv = *reinterpret_cast<const float4*>(Ls + 8 * g + 4 * h);
            if constexpr (NP == 2) {
                const float km = kmask + pexp;
                s[4 * g] = __builtin_amdgcn_exp2f(fmaf(s[4 * g], us, km - lv.x));
                s[4 * g + 1] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + 1], us, km - lv.y));
                s[4 * g + 2] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + 2], us, km - lv.z));
                s[4 * g + 3] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + 3], us, km - lv.w));
            } else {
                s[4 * g] = __builtin_amdgcn_exp2f(s[4 * g] - lv.x + kmask);
                s[4 * g + 1] = __builtin_amdgcn_exp2f(s[4 * g + 1] - lv.y + kmask);
                s[4 * g + 2] = __builtin_amdgcn_exp2f(s[4 * g + 2] - lv.z + kmask);
                s[4 * g + 3] = __builtin_amdgcn_exp2f(s[4 * g + 3] - lv.w + kmask);
            }
        }
        if constexpr (DV) {
            x6_tileT_times_acc<E, NP>(G, Gs, s, lane);
        } else {
            f32x16 dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[r] = 0.f;
            x6_rows_times_regs<E, NP>(dp, Gs, Vf, l31, h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 dv = *reinterpret_cast<const float4*>(Ds + 8 * g + 4 * h);
                if constexpr (NP == 2) {
                    const float ups = up * sds;
                    dp[4 * g] = s[4 * g] * fmaf(dp[4 * g], ups, -dv.x * sds);
                    dp[4 * g + 1] = s[4 * g + 1] * fmaf(dp[4 * g + 1], ups, -dv.y * sds);
                    dp[4 * g + 2] = s[4 * g + 2] * fmaf(dp[4 * g + 2], ups, -dv.z * sds);
                    dp[4 * g + 3] = s[4 * g + 3] * fmaf(dp[4 * g + 3], ups, -dv.w * sds);
                } else {
                    dp[4 * g] = s[4 * g] * (dp[4 * g] - dv.x);
                    dp[4 * g + 1] = s[4 * g + 1] * (dp[4 * g + 1] - dv.y);
                    dp[4 * g + 2] = s[4 * g + 2] * (dp[4 * g + 2] - dv.z);
                    dp[4 * g + 3] = s[4 * g + 3] * (dp[4 * g + 3] - dv.w);
                }
            }
            x6_tileT_times_acc<E, NP>(G, Qs, dp, lane);
        }
    }
    float am = 0.f;
    if (k_ok) {
        float* dst = (DV ? a.gv : a.gk) + b * a.gkv_sb + (long long)ki * a.g_sl + head * E;
        // NP = 2: dV holds sg 2^14 dO^T P; dK holds sq sds Q^T dS
        am = store_grad_rows<E>(dst, G, DV ? ((NP == 2) ? 6.103515625e-05f / stb : 1.f) : ((NP == 2) ? a.scale / (sta * sds) : a.scale), h);
    }
    if (a.gmax) wave_amax_to(a.gmax, am, lane);
}

// ------------------------------------------------------------------------------------------------ C ABI
template <int NP>
static int attn_fwd_x6_impl(const char* who, const float* q, const float* k, const float* v, float* out, float* lse, int B, int L,
                            int S, int H, int E, long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb,
                            long long v_sl, float scale, void* stream, const float* const* bounds = nullptr) {
    if (B <= 0 || L <= 0 || S <= 0 || H <= 0 || H > 65535 || B > 65535) {
        ign_set_error("%s: bad dimensions B=%d L=%d S=%d H=%d", who, B, L, S, H);
        return IGN_E_ARG;
    }
    if (E != 16 && E != 32 && E != 64 && E != 128) {
        ign_set_error("%s: head dimension E=%d not instantiated (16, 32, 64, 128)", who, E);
        return IGN_E_UNSUP;
    }
    if (NP == 2 && (!bounds || !bounds[0] || !bounds[1] || !bounds[2])) { ign_set_error("%s: null operand bound", who); return IGN_E_ARG; }
    const void* ptrs[5] = {q, k, v, out, lse};
    for (int i = 0; i < 5; ++i)
        if (!ptrs[i] || ((uintptr_t)ptrs[i] & 15)) {
            ign_set_error("%s: pointer %d is null or not 16-byte aligned", who, i);
            return IGN_E_ARG;
        }
    const long long st[6] = {q_sb, q_sl, k_sb, k_sl, v_sb, v_sl};
    for (int i = 0; i < 6; ++i)
        if (st[i] <= 0 || (st[i] & 3)) {
            ign_set_error("%s: stride %d = %lld must be a positive multiple of 4 elements", who, i, st[i]);
            return IGN_E_ARG;
        }
    hipStream_t s = (hipStream_t)stream;
    AttnX6Args a = {};
    a.q = q; a.k = k; a.v = v; a.out = out; a.lse_out = lse;
    a.q_sb = q_sb; a.q_sl = q_sl; a.k_sb = k_sb; a.k_sl = k_sl; a.v_sb = v_sb; a.v_sl = v_sl;
    a.B = B; a.L = L; a.S = S; a.H = H; a.E = E; a.scale = scale;
    if (NP == 2) { a.bq = bounds[0]; a.bk = bounds[1]; a.bv = bounds[2]; }
    const dim3 grid((L + 127) / 128, H, B);
    IgnScopedTimer tm("attn_fwd", s);
#define IGN_AX(EE)                                                                                                            \
    do {                                                                                                                      \
        constexpr size_t lds = (size_t)3 * AxPitch<EE>::KT * (AxPitch<EE>::K + AxPitch<EE>::V) * sizeof(unsigned short) + 256; \
        static bool once = false;                                                                                             \
        if (!once) {                                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_x6_kernel<EE, NP>),                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
            once = true;                                                                                                      \
        }                                                                                                                     \
        hipLaunchKernelGGL((attn_fwd_x6_kernel<EE, NP>), grid, dim3(256), lds, s, a);                                         \
    } while (0)
    switch (E) {
        case 16: IGN_AX(16); break;
        case 32: IGN_AX(32); break;
        case 64: IGN_AX(64); break;
        default: IGN_AX(128); break;
    }
#undef IGN_AX
    return ign_check_launch("attn_fwd_x6_kernel");
}

template <int NP>
static int attn_bwd_x6_impl(const char* who, const float* q, const float* k, const float* v, const float* out, const float* lse,
                            const float* gout, float* gq, float* gk, float* gv, float* delta_ws, int B, int L, int S, int H,
                            int E, long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb,
                            long long v_sl, float scale, void* stream, long long g_sb = 0, long long g_sl = 0,
                            const float* const* bounds = nullptr, float* g_amax = nullptr) {
    if (NP == 2 && (!bounds || !bounds[0] || !bounds[1] || !bounds[2] || !bounds[3])) {
        ign_set_error("%s: null operand bound", who);
        return IGN_E_ARG;
    }
    if (B <= 0 || L <= 0 || S <= 0 || H <= 0 || H > 65535 || B > 65535) {
        ign_set_error("%s: bad dimensions B=%d L=%d S=%d H=%d", who, B, L, S, H);
        return IGN_E_ARG;
    }
    if (E != 16 && E != 32 && E != 64 && E != 128) {
        ign_set_error("%s: head dimension E=%d not instantiated (16, 32, 64, 128)", who, E);
        return IGN_E_UNSUP;
    }
    if ((g_sb != 0) != (g_sl != 0) || g_sb < 0 || (g_sb & 3) || (g_sl & 3) || (g_sl && g_sl < (long long)H * E)) {
        ign_set_error("%s: gradient strides (%lld, %lld) must both be 0 (contiguous) or multiples of 4 with g_sl >= H*E", who, g_sb, g_sl);
        return IGN_E_ARG;
    }
    const void* ptrs[10] = {q, k, v, out, lse, gout, gq, gk, gv, delta_ws};
    for (int i = 0; i < 10; ++i)
        if (!ptrs[i] || ((uintptr_t)ptrs[i] & 15)) {
            ign_set_error("%s: pointer %d is null or not 16-byte aligned", who, i);
            return IGN_E_ARG;
        }
    const long long st[6] = {q_sb, q_sl, k_sb, k_sl, v_sb, v_sl};
    for (int i = 0; i < 6; ++i)
        if (st[i] <= 0 || (st[i] & 3)) {
            ign_set_error("%s: stride %d = %lld must be a positive multiple of 4 elements", who, i, st[i]);
            return IGN_E_ARG;
        }
    hipStream_t s = (hipStream_t)stream;
    AttnX6BwdArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = out; a.go = gout; a.lse = lse; a.delta = delta_ws; a.delta_out = delta_ws;
    a.gq = gq; a.gk = gk; a.gv = gv;
    a.q_sb = q_sb; a.q_sl = q_sl; a.k_sb = k_sb; a.k_sl = k_sl; a.v_sb = v_sb; a.v_sl = v_sl;
    a.B = B; a.L = L; a.S = S; a.H = H; a.E = E; a.scale = scale;
    if (NP == 2) { a.bq = bounds[0]; a.bk = bounds[1]; a.bv = bounds[2]; a.bg = bounds[3]; }
    a.gmax = g_amax;
    a.g_sl = g_sl ? g_sl : (long long)H * E;
    a.gq_sb = g_sb ? g_sb : (long long)L * H * E;
    a.gkv_sb = g_sb ? g_sb : (long long)S * H * E;
    const dim3 gk_grid((S + 127) / 128, H, B), gq_grid((L + 127) / 128, H, B);
#define IGN_AB(EE, KERNEL, GRID)                                                                                              \
    do {                                                                                                                      \
        static bool once = false;                                                                                             \
        if (!once) {                                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                      (int)AbCfg<EE>::LDS);                                                                   \
            once = true;                                                                                                      \
        }                                                                                                                     \
        hipLaunchKernelGGL(KERNEL, GRID, dim3(256), AbCfg<EE>::LDS, s, a);                                                    \
    } while (0)
#define IGN_AB_ALL(EE)                                                                                                        \
    do {                                                                                                                      \
        /* dQ first: it also writes delta = rowsum(dO * O), which the dK kernel reads */                                      \
        { IgnScopedTimer tm("attn_bwd_dq", s);                                                                                \
          IGN_AB(EE, (attn_bwd_dq_x6_kernel<EE, NP>), gq_grid); }                                                             \
        { IgnScopedTimer tm("attn_bwd_dkdv", s);                                                                              \
          IGN_AB(EE, (attn_bwd_dkv_x6_kernel<EE, true, NP>), gk_grid);                                                        \
          IGN_AB(EE, (attn_bwd_dkv_x6_kernel<EE, false, NP>), gk_grid); }                                                     \
    } while (0)
    switch (E) {
        case 16: IGN_AB_ALL(16); break;
        case 32: IGN_AB_ALL(32); break;
        case 64: IGN_AB_ALL(64); break;
        default: IGN_AB_ALL(128); break;
    }
#undef IGN_AB_ALL
#undef IGN_AB
    return ign_check_launch("attn_bwd_x6 kernels");
}

#define IGN_ATTN_FWD_ARGS const float* q, const float* k, const float* v, float* out, float* lse, int B, int L, int S, int H, int E, \
    long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb, long long v_sl, float scale, void* stream
#define IGN_ATTN_BWD_ARGS const float* q, const float* k, const float* v, const float* out, const float* lse, const float* gout, \
    float* gq, float* gk, float* gv, float* delta_ws, int B, int L, int S, int H, int E, long long q_sb, long long q_sl,          \
    long long k_sb, long long k_sl, long long v_sb, long long v_sl, float scale, void* stream
extern "C" int ign_attn_fwd_x6(IGN_ATTN_FWD_ARGS) {
    return attn_fwd_x6_impl<3>("ign_attn_fwd_x6", q, k, v, out, lse, B, L, S, H, E, q_sb, q_sl, k_sb, k_sl, v_sb, v_sl, scale, stream);
}
extern "C" int ign_attn_bwd_x6(IGN_ATTN_BWD_ARGS) {
    return attn_bwd_x6_impl<3>("ign_attn_bwd_x6", q, k, v, out, lse, gout, gq, gk, gv, delta_ws, B, L, S, H, E, q_sb, q_sl, k_sb, k_sl,
                               v_sb, v_sl, scale, stream);
}
// operands (Q, K, V, P, dO, dS) rounded to bf16, one product per MFMA step, fp32 accumulation and softmax: the arithmetic of the
// reference's default bf16-autocast mode for the two attention matmuls
extern "C" int ign_attn_fwd_bf16(IGN_ATTN_FWD_ARGS) {
    return attn_fwd_x6_impl<1>("ign_attn_fwd_bf16", q, k, v, out, lse, B, L, S, H, E, q_sb, q_sl, k_sb, k_sl, v_sb, v_sl, scale, stream);
}
extern "C" int ign_attn_bwd_bf16(IGN_ATTN_BWD_ARGS) {
    return attn_bwd_x6_impl<1>("ign_attn_bwd_bf16", q, k, v, out, lse, gout, gq, gk, gv, delta_ws, B, L, S, H, E, q_sb, q_sl, k_sb,
                               k_sl, v_sb, v_sl, scale, stream);
}

// Two fp16 planes, three products (include/ign_abi.h, "h3"): bq / bk / bv / bgo are device scalars holding upper bounds of
// max|q|, max|k|, max|v|, max|dO|.  g_sb = g_sl = 0: contiguous gradients; else the strided form of ign_attn_bwd_x6_strided.
extern "C" int ign_attn_fwd_h3(IGN_ATTN_FWD_ARGS, const float* bq, const float* bk, const float* bv) {
    if (E > 64) { ign_set_error("ign_attn_fwd_h3: E=%d > 64", E); return IGN_E_UNSUP; }
    const float* bounds[3] = {bq, bk, bv};
    return attn_fwd_x6_impl<2>("ign_attn_fwd_h3", q, k, v, out, lse, B, L, S, H, E, q_sb, q_sl, k_sb, k_sl, v_sb, v_sl, scale, stream, bounds);
}
extern "C" int ign_attn_bwd_h3(IGN_ATTN_BWD_ARGS, long long g_sb, long long g_sl, const float* bq, const float* bk, const float* bv,
                               const float* bgo, float* g_amax) {
    if (E > 64) { ign_set_error("ign_attn_bwd_h3: E=%d > 64", E); return IGN_E_UNSUP; }
    const float* bounds[4] = {bq, bk, bv, bgo};
    return attn_bwd_x6_impl<2>("ign_attn_bwd_h3", q, k, v, out, lse, gout, gq, gk, gv, delta_ws, B, L, S, H, E, q_sb, q_sl, k_sb, k_sl,
                               v_sb, v_sl, scale, stream, g_sb, g_sl, bounds, g_amax);
}

// The same backward writing gq / gk / gv with the caller's batch and sequence strides (elements; head stride E): lets the three
// gradients land in ONE packed (B, L, 3, H, E) buffer -- the gradient of a fused q/k/v projection -- without a gather.
extern "C" int ign_attn_bwd_x6_strided(IGN_ATTN_BWD_ARGS, long long g_sb, long long g_sl, int bf16) {
    if (g_sb <= 0 || g_sl <= 0) { ign_set_error("ign_attn_bwd_x6_strided: gradient strides must be positive"); return IGN_E_ARG; }
    return bf16 ? attn_bwd_x6_impl<1>("ign_attn_bwd_x6_strided", q, k, v, out, lse, gout, gq, gk, gv, delta_ws, B, L, S, H, E, q_sb,
                                      q_sl, k_sb, k_sl, v_sb, v_sl, scale, stream, g_sb, g_sl)
                : attn_bwd_x6_impl<3>("ign_attn_bwd_x6_strided", q, k, v, out, lse, gout, gq, gk, gv, delta_ws, B, L, S, H, E, q_sb,
                                      q_sl, k_sb, k_sl, v_sb, v_sl, scale, stream, g_sb, g_sl);
}

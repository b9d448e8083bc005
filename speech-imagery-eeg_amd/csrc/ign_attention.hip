// Fused attention core  softmax(scale * Q K^T) V  in exact fp32 on the CDNA4 matrix cores
// (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD, bitwise an fmaf chain -- no xf32/TF32 exists on gfx950).
//
// Replaces IGN/layers/SelfAttention_Family.py:56-75 (FullAttention, mask_flag=False, dropout 0) and the attention
// inside nn.TransformerEncoderLayer of IGN/model/eegcnn.py:219-228.  The reference materialises the (B,H,L,S)
// scores (8.2 GB per layer at B=256, H=8, L=S=1000); these kernels keep one 32x32 score tile per wavefront in
// accumulator registers (flash-style online softmax) and save only the log-sum-exp per query for the backward.
//
// Orientation.  Every product is arranged so that the index that is REDUCED next sits in the accumulator
// registers and the index that is KEPT sits on the lane (cdna_hip_programming.md, "accumulator tile as the next
// MFMA's operand"): the 16 accumulator registers of lane (c = lane&31, h = lane>>5) hold rows
// r -> (r&3) + 8*(r>>2) + 4*h of column c.
//   forward / dQ kernel: S^T = K Q^T  (rows = keys, column = query on the lane): the softmax row reduction is 16
//       in-register values + one lane^32 exchange, and P^T is already the B operand of O^T += V^T P^T.
//   dK/dV kernel:        S = Q K^T    (rows = queries, column = key on the lane): P and dS are already the B
//       operands of dV^T += dO^T P and dK^T += Q^T dS.
// The k index of a 32x32x2 step is split over the lane halves: half h covers e in [h*E/2, (h+1)*E/2), so a lane's
// operand stream is E/2 CONTIGUOUS floats (ds_read_b128), not a stride-2 gather.
#include "ign_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnArgs {
    const float *q, *k, *v;      // (B,L,H,E) / (B,S,H,E) with element strides sb (batch), sl (sequence); head stride E
    const float *o, *lse, *go, *delta;
    float *out, *lse_out, *gq, *gk, *gv, *delta_out;
    long long q_sb, q_sl, k_sb, k_sl, v_sb, v_sl;
    int B, L, S, H, E;
    float scale;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

constexpr int ATT_KT = 64;      // rows staged per LDS tile

// cooperative load of `rows` x E floats (row r at src + r*stride) into dst[r*PITCH + e]; rows >= nvalid zero-filled
template <int E, int PITCH>
__device__ __forceinline__ void stage_tile(float* dst, const float* src, long long stride, int nvalid, int rows) {
    constexpr int V4 = E / 4;
    for (int i = threadIdx.x; i < rows * V4; i += blockDim.x) {
        const int r = i / V4, c4 = (i - r * V4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < nvalid) v = *reinterpret_cast<const float4*>(src + (long long)r * stride + c4);
        *reinterpret_cast<float4*>(dst + r * PITCH + c4) = v;
    }
}

// ------------------------------------------------------------------------------------------------ forward
template <int E>
__global__ void __launch_bounds__(256) attn_fwd_kernel(const AttnArgs a) {
    constexpr int EH = E / 2, ED = (E + 31) / 32, PITCH = E + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + ATT_KT * PITCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int qi = blockIdx.x * 128 + wave * 32 + l31;
    const bool q_ok = qi < a.L;

    float Qf[EH];                                       // Q[qi][h*EH + kk] * scale
    {
        const float* qp = a.q + b * a.q_sb + (long long)(q_ok ? qi : a.L - 1) * a.q_sl + head * E + h * EH;
#pragma unroll
        for (int kk = 0; kk < EH; kk += 4) {
            const float4 t = *reinterpret_cast<const float4*>(qp + kk);
            Qf[kk] = t.x * a.scale; Qf[kk + 1] = t.y * a.scale; Qf[kk + 2] = t.z * a.scale; Qf[kk + 3] = t.w * a.scale;
        }
    }
    f32x16 O[ED];
#pragma unroll
    for (int d = 0; d < ED; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    const float* kbase = a.k + b * a.k_sb + head * E;
    const float* vbase = a.v + b * a.v_sb + head * E;
    for (int kt0 = 0; kt0 < a.S; kt0 += ATT_KT) {
        __syncthreads();
        stage_tile<E, PITCH>(Ks, kbase + (long long)kt0 * a.k_sl, a.k_sl, a.S - kt0, ATT_KT);
        stage_tile<E, PITCH>(Vs, vbase + (long long)kt0 * a.v_sl, a.v_sl, a.S - kt0, ATT_KT);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < ATT_KT / 32; ++sub) {
            const int kb = sub * 32;
            if (kt0 + kb < a.S) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float* kr = Ks + (kb + l31) * PITCH + h * EH;
#pragma unroll
                for (int kk = 0; kk < EH; kk += 4) {
                    const float4 kv = *reinterpret_cast<const float4*>(kr + kk);
                    acc = MFMA(kv.x, Qf[kk], acc);
                    acc = MFMA(kv.y, Qf[kk + 1], acc);
                    acc = MFMA(kv.z, Qf[kk + 2], acc);
                    acc = MFMA(kv.w, Qf[kk + 3], acc);
                }
                // online softmax over this lane's 16 keys + the partner half's 16 keys
                float mloc = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (kt0 + kb + acc_row(r, h) >= a.S) acc[r] = -INFINITY;
                    mloc = fmaxf(mloc, acc[r]);
                }
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                const float mnew = fmaxf(m, mloc);
                const float alpha = __expf(m - mnew);
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[r] = __expf(acc[r] - mnew);
                    psum += acc[r];
                }
                psum += __shfl_xor(psum, 32, 64);
                l = l * alpha + psum;
                m = mnew;
#pragma unroll
                for (int d = 0; d < ED; ++d) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
                    const bool d_ok = d * 32 + l31 < E;
                    const float* vr = Vs + kb * PITCH + d * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float vv = d_ok ? vr[acc_row(r, h) * PITCH] : 0.f;
                        O[d] = MFMA(vv, acc[r], O[d]);
                    }
                }
            }
        }
    }
    if (q_ok) {
        const float inv = 1.f / l;
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
        if (h == 0) a.lse_out[((long long)b * a.H + head) * a.L + qi] = m + __logf(l);
    }
}

// ------------------------------------------------------------------------------------------------ delta = rowsum(dO * O)
__global__ void __launch_bounds__(256) attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ go,
                                                         float* __restrict__ delta, int B, int L, int H, int E) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;      // over (b, l, h)
    if (idx >= (long long)B * L * H) return;
    const float* po = o + idx * E;
    const float* pg = go + idx * E;
    float s = 0.f;
    for (int e = 0; e < E; e += 4) {
        const float4 x = *reinterpret_cast<const float4*>(po + e);
        const float4 y = *reinterpret_cast<const float4*>(pg + e);
        s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    const int hh = (int)(idx % H);
    const long long bl = idx / H;
    const int li = (int)(bl % L);
    const int bb = (int)(bl / L);
    delta[((long long)bb * H + hh) * L + li] = s;
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// Block = 4 waves x 32 keys; loops over query tiles.  Per (32 queries x 32 keys): S = Q K^T, dP = dO V^T,
// P = exp(S - lse), dS = P (dP - delta) scale, dV^T += dO^T P, dK^T += Q^T dS.
template <int E>
__global__ void __launch_bounds__(256, 2) attn_bwd_dkdv_kernel(const AttnArgs a) {
    constexpr int EH = E / 2, ED = (E + 31) / 32, PITCH = E + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                                   // [ATT_KT][PITCH]
    float* Gs = smem + ATT_KT * PITCH;                  // dO tile
    float* Ls = smem + 2 * ATT_KT * PITCH;              // lse[ATT_KT], delta[ATT_KT]
    float* Ds = Ls + ATT_KT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int ki = blockIdx.x * 128 + wave * 32 + l31;  // this lane's key
    const bool k_ok = ki < a.S;

    float Kf[EH], Vf[EH];                               // K[ki][h*EH+kk] (pre-scaled), V[ki][h*EH+kk]
    {
        const long long row = k_ok ? ki : a.S - 1;
        const float* kp = a.k + b * a.k_sb + row * a.k_sl + head * E + h * EH;
        const float* vp = a.v + b * a.v_sb + row * a.v_sl + head * E + h * EH;
#pragma unroll
        for (int kk = 0; kk < EH; kk += 4) {
            const float4 t = *reinterpret_cast<const float4*>(kp + kk);
            const float4 u = *reinterpret_cast<const float4*>(vp + kk);
            Kf[kk] = t.x * a.scale; Kf[kk + 1] = t.y * a.scale; Kf[kk + 2] = t.z * a.scale; Kf[kk + 3] = t.w * a.scale;
            Vf[kk] = u.x; Vf[kk + 1] = u.y; Vf[kk + 2] = u.z; Vf[kk + 3] = u.w;
        }
    }
    f32x16 dV[ED], dK[ED];
#pragma unroll
    for (int d = 0; d < ED; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dV[d][r] = 0.f; dK[d][r] = 0.f; }

    const float* qbase = a.q + b * a.q_sb + head * E;
    const float* gbase = a.go + (long long)b * a.L * a.H * E + head * E;
    const long long g_sl = (long long)a.H * E;
    const float* lse_b = a.lse + ((long long)b * a.H + head) * a.L;
    const float* del_b = a.delta + ((long long)b * a.H + head) * a.L;

    for (int qt0 = 0; qt0 < a.L; qt0 += ATT_KT) {
        __syncthreads();
        stage_tile<E, PITCH>(Qs, qbase + (long long)qt0 * a.q_sl, a.q_sl, a.L - qt0, ATT_KT);
        stage_tile<E, PITCH>(Gs, gbase + (long long)qt0 * g_sl, g_sl, a.L - qt0, ATT_KT);
        if (threadIdx.x < ATT_KT) {
            const int q = qt0 + threadIdx.x;
            Ls[threadIdx.x] = q < a.L ? lse_b[q] : INFINITY;       // exp(s - inf) = 0: padded queries contribute nothing
            Ds[threadIdx.x] = q < a.L ? del_b[q] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < ATT_KT / 32; ++sub) {
            const int qb = sub * 32;
            if (qt0 + qb < a.L) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                const float* qr = Qs + (qb + l31) * PITCH + h * EH;
                const float* gr = Gs + (qb + l31) * PITCH + h * EH;
#pragma unroll
                for (int kk = 0; kk < EH; kk += 4) {
                    const float4 qv = *reinterpret_cast<const float4*>(qr + kk);
                    const float4 gv = *reinterpret_cast<const float4*>(gr + kk);
                    s = MFMA(qv.x, Kf[kk], s);      dp = MFMA(gv.x, Vf[kk], dp);
                    s = MFMA(qv.y, Kf[kk + 1], s);  dp = MFMA(gv.y, Vf[kk + 1], dp);
                    s = MFMA(qv.z, Kf[kk + 2], s);  dp = MFMA(gv.z, Vf[kk + 2], dp);
                    s = MFMA(qv.w, Kf[kk + 3], s);  dp = MFMA(gv.w, Vf[kk + 3], dp);
                }
                // rows = queries qb + acc_row(r,h); column = this lane's key
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qq = qb + acc_row(r, h);
                    const float p = k_ok ? __expf(s[r] - Ls[qq]) : 0.f;
                    s[r] = p;
                    dp[r] = p * (dp[r] - Ds[qq]) * a.scale;
                }
#pragma unroll
                for (int d = 0; d < ED; ++d) {
                    const bool d_ok = d * 32 + l31 < E;
                    const float* gc = Gs + qb * PITCH + d * 32 + l31;
                    const float* qc = Qs + qb * PITCH + d * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int off = acc_row(r, h) * PITCH;
                        const float gvv = d_ok ? gc[off] : 0.f;
                        const float qvv = d_ok ? qc[off] : 0.f;
                        dV[d] = MFMA(gvv, s[r], dV[d]);
                        dK[d] = MFMA(qvv, dp[r], dK[d]);
                    }
                }
            }
        }
    }
    if (k_ok) {
        float* pk = a.gk + (((long long)b * a.S + ki) * a.H + head) * E;
        float* pv = a.gv + (((long long)b * a.S + ki) * a.H + head) * E;
#pragma unroll
        for (int d = 0; d < ED; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * h;
                if (d0 < E) {
                    *reinterpret_cast<float4*>(pk + d0) = make_float4(dK[d][4 * g], dK[d][4 * g + 1], dK[d][4 * g + 2], dK[d][4 * g + 3]);
                    *reinterpret_cast<float4*>(pv + d0) = make_float4(dV[d][4 * g], dV[d][4 * g + 1], dV[d][4 * g + 2], dV[d][4 * g + 3]);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// Block = 4 waves x 32 queries; loops over key tiles.  S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - delta) scale,
// dQ^T += K^T dS^T.  Recomputing S and dP here (7 products in total instead of 5) keeps dQ free of float atomics:
// the result is bitwise reproducible.
template <int E>
__global__ void __launch_bounds__(256) attn_bwd_dq_kernel(const AttnArgs a) {
    constexpr int EH = E / 2, ED = (E + 31) / 32, PITCH = E + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;
    float* Vs = smem + ATT_KT * PITCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    const int qi = blockIdx.x * 128 + wave * 32 + l31;
    const bool q_ok = qi < a.L;
    const long long qrow = q_ok ? qi : a.L - 1;

    float Qf[EH], Gf[EH];
    {
        const float* qp = a.q + b * a.q_sb + qrow * a.q_sl + head * E + h * EH;
        const float* gp = a.go + (((long long)b * a.L + qrow) * a.H + head) * E + h * EH;
#pragma unroll
        for (int kk = 0; kk < EH; kk += 4) {
            const float4 t = *reinterpret_cast<const float4*>(qp + kk);
            const float4 u = *reinterpret_cast<const float4*>(gp + kk);
            Qf[kk] = t.x * a.scale; Qf[kk + 1] = t.y * a.scale; Qf[kk + 2] = t.z * a.scale; Qf[kk + 3] = t.w * a.scale;
            Gf[kk] = u.x; Gf[kk + 1] = u.y; Gf[kk + 2] = u.z; Gf[kk + 3] = u.w;
        }
    }
    const float lse = a.lse[((long long)b * a.H + head) * a.L + qrow];
    const float delta = a.delta[((long long)b * a.H + head) * a.L + qrow];
    f32x16 dQ[ED];
#pragma unroll
    for (int d = 0; d < ED; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQ[d][r] = 0.f;

    const float* kbase = a.k + b * a.k_sb + head * E;
    const float* vbase = a.v + b * a.v_sb + head * E;
    for (int kt0 = 0; kt0 < a.S; kt0 += ATT_KT) {
        __syncthreads();
        stage_tile<E, PITCH>(Ks, kbase + (long long)kt0 * a.k_sl, a.k_sl, a.S - kt0, ATT_KT);
        stage_tile<E, PITCH>(Vs, vbase + (long long)kt0 * a.v_sl, a.v_sl, a.S - kt0, ATT_KT);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < ATT_KT / 32; ++sub) {
            const int kb = sub * 32;
            if (kt0 + kb < a.S) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                const float* kr = Ks + (kb + l31) * PITCH + h * EH;
                const float* vr = Vs + (kb + l31) * PITCH + h * EH;
#pragma unroll
                for (int kk = 0; kk < EH; kk += 4) {
                    const float4 kv = *reinterpret_cast<const float4*>(kr + kk);
                    const float4 vv = *reinterpret_cast<const float4*>(vr + kk);
                    s = MFMA(kv.x, Qf[kk], s);      dp = MFMA(vv.x, Gf[kk], dp);
                    s = MFMA(kv.y, Qf[kk + 1], s);  dp = MFMA(vv.y, Gf[kk + 1], dp);
                    s = MFMA(kv.z, Qf[kk + 2], s);  dp = MFMA(vv.z, Gf[kk + 2], dp);
                    s = MFMA(kv.w, Qf[kk + 3], s);  dp = MFMA(vv.w, Gf[kk + 3], dp);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const bool key_ok = kt0 + kb + acc_row(r, h) < a.S;
                    const float p = key_ok ? __expf(s[r] - lse) : 0.f;
                    dp[r] = p * (dp[r] - delta) * a.scale;
                }
#pragma unroll
                for (int d = 0; d < ED; ++d) {
                    const bool d_ok = d * 32 + l31 < E;
                    const float* kc = Ks + kb * PITCH + d * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float kvv = d_ok ? kc[acc_row(r, h) * PITCH] : 0.f;
                        dQ[d] = MFMA(kvv, dp[r], dQ[d]);
                    }
                }
            }
        }
    }
    if (q_ok) {
        float* pq = a.gq + (((long long)b * a.L + qi) * a.H + head) * E;
#pragma unroll
        for (int d = 0; d < ED; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * h;
                if (d0 < E)
                    *reinterpret_cast<float4*>(pq + d0) = make_float4(dQ[d][4 * g], dQ[d][4 * g + 1], dQ[d][4 * g + 2], dQ[d][4 * g + 3]);
            }
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
static int attn_check(const char* who, int B, int L, int S, int H, int E, const long long* strides, const void* const* ptrs,
                      int nptr) {
    if (B <= 0 || L <= 0 || S <= 0 || H <= 0) {
        ign_set_error("%s: bad dimensions B=%d L=%d S=%d H=%d", who, B, L, S, H);
        return IGN_E_ARG;
    }
    if (E != 16 && E != 32 && E != 64 && E != 128) {
        ign_set_error("%s: head dimension E=%d not instantiated (16, 32, 64, 128)", who, E);
        return IGN_E_UNSUP;
    }
    for (int i = 0; i < nptr; ++i)
        if (!ptrs[i] || ((uintptr_t)ptrs[i] & 15)) {
            ign_set_error("%s: pointer %d is null or not 16-byte aligned", who, i);
            return IGN_E_ARG;
        }
    for (int i = 0; i < 6; ++i)
        if (strides[i] <= 0 || (strides[i] & 3)) {
            ign_set_error("%s: stride %d = %lld must be a positive multiple of 4 elements", who, i, strides[i]);
            return IGN_E_ARG;
        }
    if (H > 65535 || B > 65535) {
        ign_set_error("%s: H and B are grid dimensions (<= 65535)", who);
        return IGN_E_ARG;
    }
    return 0;
}

#define ATTN_DISPATCH(E_, KERNEL, grid, lds, stream, args)                                           \
    switch (E_) {                                                                                   \
        case 16: hipLaunchKernelGGL((KERNEL<16>), grid, dim3(256), lds, stream, args); break;       \
        case 32: hipLaunchKernelGGL((KERNEL<32>), grid, dim3(256), lds, stream, args); break;       \
        case 64: hipLaunchKernelGGL((KERNEL<64>), grid, dim3(256), lds, stream, args); break;       \
        default: hipLaunchKernelGGL((KERNEL<128>), grid, dim3(256), lds, stream, args); break;      \
    }

extern "C" int ign_attn_fwd(const float* q, const float* k, const float* v, float* out, float* lse, int B, int L, int S,
                            int H, int E, long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb,
                            long long v_sl, float scale, void* stream) {
    static const char* who = "ign_attn_fwd";
    const long long st[6] = {q_sb, q_sl, k_sb, k_sl, v_sb, v_sl};
    const void* ptrs[5] = {q, k, v, out, lse};
    int rc;
    if ((rc = attn_check(who, B, L, S, H, E, st, ptrs, 5))) return rc;
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.out = out; a.lse_out = lse;
    a.q_sb = q_sb; a.q_sl = q_sl; a.k_sb = k_sb; a.k_sl = k_sl; a.v_sb = v_sb; a.v_sl = v_sl;
    a.B = B; a.L = L; a.S = S; a.H = H; a.E = E; a.scale = scale;
    const size_t lds = (size_t)2 * ATT_KT * (E + 4) * sizeof(float);
    const dim3 grid((L + 127) / 128, H, B);
    IgnScopedTimer tm("attn_fwd", (hipStream_t)stream);
    ATTN_DISPATCH(E, attn_fwd_kernel, grid, lds, (hipStream_t)stream, a);
    return ign_check_launch("attn_fwd_kernel");
}

extern "C" int ign_attn_bwd(const float* q, const float* k, const float* v, const float* out, const float* lse,
                            const float* gout, float* gq, float* gk, float* gv, float* delta_ws, int B, int L, int S, int H,
                            int E, long long q_sb, long long q_sl, long long k_sb, long long k_sl, long long v_sb,
                            long long v_sl, float scale, void* stream) {
    static const char* who = "ign_attn_bwd";
    const long long st[6] = {q_sb, q_sl, k_sb, k_sl, v_sb, v_sl};
    const void* ptrs[10] = {q, k, v, out, lse, gout, gq, gk, gv, delta_ws};
    int rc;
    if ((rc = attn_check(who, B, L, S, H, E, st, ptrs, 10))) return rc;
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.o = out; a.lse = lse; a.go = gout; a.delta = delta_ws;
    a.gq = gq; a.gk = gk; a.gv = gv; a.delta_out = delta_ws;
    a.q_sb = q_sb; a.q_sl = q_sl; a.k_sb = k_sb; a.k_sl = k_sl; a.v_sb = v_sb; a.v_sl = v_sl;
    a.B = B; a.L = L; a.S = S; a.H = H; a.E = E; a.scale = scale;
    hipStream_t s = (hipStream_t)stream;
    {
        const long long n = (long long)B * L * H;
        IgnScopedTimer tm("attn_delta", s);
        hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, gout, delta_ws, B, L, H, E);
    }
    if ((rc = ign_check_launch("attn_delta_kernel"))) return rc;
    {
        const size_t lds = ((size_t)2 * ATT_KT * (E + 4) + 2 * ATT_KT) * sizeof(float);
        const dim3 grid((S + 127) / 128, H, B);
        IgnScopedTimer tm("attn_bwd_dkdv", s);
        ATTN_DISPATCH(E, attn_bwd_dkdv_kernel, grid, lds, s, a);
    }
    if ((rc = ign_check_launch("attn_bwd_dkdv_kernel"))) return rc;
    {
        const size_t lds = (size_t)2 * ATT_KT * (E + 4) * sizeof(float);
        const dim3 grid((L + 127) / 128, H, B);
        IgnScopedTimer tm("attn_bwd_dq", s);
        ATTN_DISPATCH(E, attn_bwd_dq_kernel, grid, lds, s, a);
    }
    return ign_check_launch("attn_bwd_dq_kernel");
}

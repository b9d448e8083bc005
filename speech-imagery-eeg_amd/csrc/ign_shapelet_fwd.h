// Shapelet forward: sliding-window distance + gate, one wavefront per (batch row, channel).
//
// Replaces IGN/model/Shapelet.py:60-84 (and :96-111 for the LTS gate) without materialising the
// (B,Tw,K,C,L) broadcast.  Bound: fp32 VALU (|x-w| accumulate is not a contraction) -- 2 VALU ops per
// (b,c,k,t,j) element; memory traffic is ~1e3x below the HBM roofline, so the design goal is an inner
// loop that is nothing but `v_sub_f32 ; v_add_f32 |.|`:
//   * lane l of the wave owns TT consecutive windows t = l*TT .. l*TT+TT-1 (TT = ceil(Tw/64): the whole
//     row fits one wave pass, so the row reductions (arg-max, soft-max sums, min) are wave shuffles);
//   * the x row is staged once in LDS (4 KB); each lane keeps a sliding register window of TT+J-1
//     samples, so every LDS word read feeds TT*KT subtract/accumulate pairs;
//   * all lanes share the channel, so w[k,c,j] is wave-uniform: it arrives through scalar loads and
//     is consumed as the SGPR operand of v_sub -- no VGPR, no LDS traffic;
//   * KT shapelets share each x register (KT*TT accumulators per lane).
#pragma once
#include "ign_common.h"

// Up to this many windows per lane the shapelet values of a step are all requested before anything else is scheduled (one exposed
// scalar-cache latency per step); beyond it the pinned order costs registers -> scratch (same box: L = 100 0.638 -> 0.676 ms at 16,
// L = 300 1.136 -> 1.11 ms at 12 against 8).
#ifndef IGN_FWD_BARRIER_TT
#define IGN_FWD_BARRIER_TT 12
#endif
template <int TT> struct FwdJ { static constexpr int J = (TT <= 8) ? 8 : 4; };

// Wave reductions on the DPP path (pure VALU): two quad permutes, row_half_mirror, row_mirror, row_bcast:15, row_bcast:31.
// After the four in-row steps every lane of a 16-lane row holds the row's result; the two broadcast steps leave the
// 64-lane result in the last row, i.e. in LANE 63 -- the only lane that writes the row's outputs.  `__shfl_xor` compiles
// to ds_bpermute_b32: the 30 dependent LDS round trips per shapelet it cost were ~20 % of the epilogue (s_memtime stamps).
#define IGN_DPP(v, ctrl, rmask, ident) \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(ident)), __builtin_bit_cast(int, (float)(v)), (ctrl), (rmask), 0xf, false))
#define IGN_DPPI(v, ctrl, rmask, ident) __builtin_amdgcn_update_dpp((int)(ident), (int)(v), (ctrl), (rmask), 0xf, false)
constexpr int DPP_QP_XOR1 = 0xB1, DPP_QP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_BCAST15 = 0x142,
              DPP_BCAST31 = 0x143;

__device__ __forceinline__ float wave_sum_l63(float v) {
    v += IGN_DPP(v, DPP_QP_XOR1, 0xf, 0.f);
    v += IGN_DPP(v, DPP_QP_XOR2, 0xf, 0.f);
    v += IGN_DPP(v, DPP_HALF_MIRROR, 0xf, 0.f);
    v += IGN_DPP(v, DPP_MIRROR, 0xf, 0.f);
    v += IGN_DPP(v, DPP_BCAST15, 0xa, 0.f);
    v += IGN_DPP(v, DPP_BCAST31, 0xc, 0.f);
    return v;                                   // valid in lane 63
}
__device__ __forceinline__ float wave_min_l63(float v) {
    v = fminf(v, IGN_DPP(v, DPP_QP_XOR1, 0xf, INFINITY));
    v = fminf(v, IGN_DPP(v, DPP_QP_XOR2, 0xf, INFINITY));
    v = fminf(v, IGN_DPP(v, DPP_HALF_MIRROR, 0xf, INFINITY));
    v = fminf(v, IGN_DPP(v, DPP_MIRROR, 0xf, INFINITY));
    v = fminf(v, IGN_DPP(v, DPP_BCAST15, 0xa, INFINITY));
    v = fminf(v, IGN_DPP(v, DPP_BCAST31, 0xc, INFINITY));
    return v;                                   // valid in lane 63
}
// arg-max with "first index wins" on ties: (best, idx) <- better of (best, idx) and the DPP partner
#define IGN_ARGMAX_STEP(ctrl, rmask)                                                        \
    do {                                                                                    \
        const float ob = IGN_DPP(best, ctrl, rmask, -INFINITY);                             \
        const int oi = IGN_DPPI(idx, ctrl, rmask, 0x7fffffff);                              \
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }                 \
    } while (0)
__device__ __forceinline__ float wave_bcast_l63(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The body of one block; (bx, by) play the role of blockIdx.x / .y (kept as parameters: round 2's one-grid bank kernel called the
// same body with a remapped block index, measured slower and was removed -- DESIGN 4.1b).  The calling kernel's ONLY argument must be
// the ShpFwdArgs block `a`: the epilogue re-reads it from offset 0 of the kernarg segment.
template <int TT, int KT, int DIST>
__device__ __forceinline__ void shp_fwd_body(const ShpFwdArgs& a, const int bx, const int by, float* smem) {
    constexpr int J = FwdJ<TT>::J;
    const int wpb = blockDim.x >> 6;
    // the wave index is uniform but the compiler cannot know: without readfirstlane every row / output address below is per-lane
    // 64-bit VALU arithmetic (31 address pairs, formed before the distance loop and spilled: 248 B/lane of scratch at TT = 15)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int nbg = (a.B + wpb - 1) / wpb;
    const int c = bx / nbg;                         // block-uniform: every wave of the block shares w[:,c,:]
    const int bg = bx - c * nbg;
    int b = bg * wpb + wave;
    const bool row_ok = b < a.B;
    if (!row_ok) b = a.B - 1;
    const int k0 = a.k0 + by * KT;

    float* xs = smem + wave * a.xs_len;
    {
        // batches of 8 loads in flight before the first LDS store: one memory round trip per 512 samples instead of one
        // per 64 (the backward's staging showed ~1.2k cycles per serialised round trip under load)
        const float* row = a.xn + ((size_t)b * a.C + c) * a.T;
        for (int i0 = lane; i0 < a.xs_len; i0 += 8 * 64) {
            float xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u;
                xv[u] = (i < a.T) ? row[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + 64 * u;
                if (i < a.xs_len) xs[i] = xv[u];
            }
        }
    }
    __syncthreads();

    const size_t wks = (size_t)a.C * a.L;
    // w is read-only for the whole launch and its address is wave-uniform: read it through the constant
    // address space so the loads are s_load_dword* (scalar cache) and each value is an SGPR operand.
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    const cfloat_p wk = (cfloat_p)(uintptr_t)(a.w + ((size_t)k0 * a.C + c) * a.L);
    const int L = a.L;

    // Row statistics of this lane (merged across lanes at the end).  They are NOT kept in registers across
    // the distance loop: rows longer than 64*TT windows (npass > 1, not the case for any T <= 1024+L)
    // park them in LDS between passes, so the hot loop's register budget is acc + window only.
    float* park = smem + (blockDim.x >> 6) * a.xs_len + threadIdx.x;    // [5*KT][blockDim.x], npass > 1 only

    for (int pass = 0; pass < a.npass; ++pass) {
        const int tl = (pass * 64 + lane) * TT;     // first window owned by this lane
        const float* xl = xs + tl * a.stride;       // stride != 1 only with TT == 1
        float acc[KT][TT];
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int t = 0; t < TT; ++t) acc[k][t] = 0.f;
        float xw[TT + J - 1];
#pragma unroll
        for (int i = 0; i < TT - 1; ++i) xw[i] = xl[i];
        // cosine / pearson (Shapelet.py:64-69): the distance is 1 - <x_win, w> / (norms); the window norms are
        // accumulated beside the dot products (1/KT extra work), the shapelet norms from the wave-uniform weights.
        float xsq[DIST >= DIST_COS ? TT : 1], xsm[DIST == DIST_PEARSON ? TT : 1], wn2[DIST >= DIST_COS ? KT : 1];
        if (DIST >= DIST_COS) {
#pragma unroll
            for (int t = 0; t < TT; ++t) { xsq[t] = 0.f; if (DIST == DIST_PEARSON) xsm[t] = 0.f; }
#pragma unroll
            for (int k = 0; k < KT; ++k) wn2[k] = 0.f;
        }

        int j0 = 0;
        for (; j0 + J <= L; j0 += J) {
#pragma unroll
            for (int jj = 0; jj < J; ++jj) xw[TT - 1 + jj] = xl[j0 + TT - 1 + jj];
            // all KT*J shapelet values of this step are requested up front (s_load, wave-uniform); with the barrier nothing is
            // scheduled across: one exposed scalar-cache latency per step instead of one per shapelet
            float wvs[J][KT];
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int jj = 0; jj < J; ++jj) wvs[jj][k] = wk[k * wks + j0 + jj];
            if constexpr (TT <= IGN_FWD_BARRIER_TT) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < J; ++jj) {
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const float wv = wvs[jj][k];                     // SGPR operand
                    if (DIST >= DIST_COS) wn2[k] = fmaf(wv, wv, wn2[k]);
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        if (DIST >= DIST_COS) {
                            acc[k][t] = fmaf(xw[t + jj], wv, acc[k][t]);         // sliding dot product
                        } else {
                            const float df = xw[t + jj] - wv;
                            if (DIST == DIST_L1) acc[k][t] += fabsf(df);
                            else                 acc[k][t] = fmaf(df, df, acc[k][t]);
                        }
                    }
                }
                if (DIST >= DIST_COS) {
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        xsq[t] = fmaf(xw[t + jj], xw[t + jj], xsq[t]);
                        if (DIST == DIST_PEARSON) xsm[t] += xw[t + jj];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TT - 1; ++i) xw[i] = xw[i + J];
        }
        for (; j0 < L; ++j0) {                      // L % J tail, one sample at a time
            xw[TT - 1] = xl[j0 + TT - 1];
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const float wv = wk[k * wks + j0];
                if (DIST >= DIST_COS) wn2[k] = fmaf(wv, wv, wn2[k]);
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    if (DIST >= DIST_COS) {
                        acc[k][t] = fmaf(xw[t], wv, acc[k][t]);
                    } else {
                        const float df = xw[t] - wv;
                        if (DIST == DIST_L1) acc[k][t] += fabsf(df);
                        else                 acc[k][t] = fmaf(df, df, acc[k][t]);
                    }
                }
            }
            if (DIST >= DIST_COS) {
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    xsq[t] = fmaf(xw[t], xw[t], xsq[t]);
                    if (DIST == DIST_PEARSON) xsm[t] += xw[t];
                }
            }
#pragma unroll
            for (int i = 0; i < TT - 1; ++i) xw[i] = xw[i + 1];
        }

        // ---- per-pass epilogue, one shapelet at a time so only ONE set of row statistics is live:
        // d = mean -> gate statistics -> coalesced store of d -> (last pass) merge of the 64 lanes and outputs.
        // Branch-free statistics: window positions past the end of the row (only in the last lanes) get d = +BIG, so
        // p = exp(-(eps BIG)^2) = 0 and exp(-(BIG - m)) = 0 never win the arg-max / arg-min and add nothing to M; the
        // RBF soft-max weight exp(p) of such a slot is exactly 1, which is subtracted from Z afterwards.
        // Everything the epilogue addresses is wave-uniform.  Two opaque scalar copies keep that arithmetic HERE: the row index
        // (so that the 5 x 6 output addresses are not formed before the distance loop and carried across it) and the argument
        // block, re-read from the kernarg segment at the point of use (`a` is the kernel's only argument, at offset 0) instead
        // of holding ~30 SGPRs across the loop.  Together with the scalar wave index above this removed all scratch: the
        // hoisted per-lane 64-bit addresses used to be spilled, 31 pairs per block = 1 GB/step of scratch writes in WRITE_SIZE.
        int be = b;
        asm volatile("" : "+s"(be));
        typedef const __attribute__((address_space(4))) ShpFwdArgs* kargs_p;
        kargs_p ae = (kargs_p)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ae));
        const int nvalid = min(TT, max(0, ae->Tw - tl));
        const bool lds_store = (ae->npass == 1) && ae->d;          // x row no longer needed: reuse its LDS as a transpose buffer
        const bool last_pass = pass + 1 == ae->npass;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            float rb, rd, rZ, rM;
            int ri;
            if (pass == 0) {
                rb = -INFINITY; rd = INFINITY; rZ = 0.f; rM = 0.f; ri = 0x7fffffff;
            } else {
                rb = park[(5 * k + 0) * blockDim.x]; rd = park[(5 * k + 1) * blockDim.x];
                rZ = park[(5 * k + 2) * blockDim.x]; rM = park[(5 * k + 3) * blockDim.x];
                ri = __float_as_int(park[(5 * k + 4) * blockDim.x]);
            }
            float* drow = ae->d ? ae->d + (((size_t)be * ae->C + c) * ae->K + (k0 + k)) * ae->Tw : nullptr;
            float dv[TT];
            if (DIST >= DIST_COS) {
                // cosine:  d = 1 - dot / (max(|x|,1e-8) max(|w|,1e-8))              (F.cosine_similarity)
                // pearson: d = 1 - dot_c / (sqrt(vx vw) + 1e-8), w already centred by the caller, vx = sum x^2 - (sum x)^2/L
                const float wn = sqrtf(wn2[k]);
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    float xn;
                    if (DIST == DIST_COS) xn = sqrtf(xsq[t]);
                    else                  xn = sqrtf(fmaxf(xsq[t] - xsm[t] * xsm[t] * ae->invL, 0.f));
                    const float den = (DIST == DIST_COS) ? fmaxf(xn, 1e-8f) * fmaxf(wn, 1e-8f) : xn * wn + 1e-8f;
                    dv[t] = (t < nvalid) ? 1.f - acc[k][t] / den : 1e18f;
                    if (k == 0 && ae->xstat && row_ok && t < nvalid) ae->xstat[((size_t)be * ae->C + c) * ae->Tw + tl + t] = xn;
                }
            } else {
#pragma unroll
                for (int t = 0; t < TT; ++t) dv[t] = (t < nvalid) ? acc[k][t] * ae->invL : 1e18f;
            }
            if (ae->gate == GATE_RBF) {
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    const float u = ae->eps * dv[t];
                    const float p = __expf(-(u * u));
                    const float e = __expf(p);
                    rZ += e;
                    rM = fmaf(e, p, rM);
                    if (p > rb) { rb = p; ri = tl + t; }
                    rd = fminf(rd, dv[t]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the inlined exp bodies from interleaving (VGPR pressure)
                }
                rZ -= (float)(TT - nvalid);
            } else {                                  // LTS: soft-min over d, stabilised by the running min
                float pmin = dv[0];
#pragma unroll
                for (int t = 1; t < TT; ++t) pmin = fminf(pmin, dv[t]);
                // rb holds -(running min) so the arg-min tie rule is "first index"
                const float mold = rd;
                const float mnew = fminf(mold, pmin);
                const float sc = (mold < INFINITY) ? __expf(mnew - mold) : 0.f;
                rZ *= sc; rM *= sc;
                if (mnew < 1e17f) {
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        const float e = __expf(mnew - dv[t]);
                        rZ += e;
                        rM = fmaf(e, dv[t], rM);
                        if (-dv[t] > rb) { rb = -dv[t]; ri = tl + t; }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    rd = mnew;
                }
            }
            if (drow) {
                if (lds_store) {
                    // lane-major registers -> time-major LDS -> coalesced 256-B global stores
                    __syncthreads();
#pragma unroll
                    for (int t = 0; t < TT; ++t) xs[lane * TT + t] = dv[t];
                    __syncthreads();
                    if (row_ok)
#pragma unroll
                        for (int i = 0; i < TT; ++i) {
                            const int idx = lane + 64 * i;
                            if (idx < ae->Tw) drow[(unsigned)idx] = xs[idx];
                        }
                } else if (row_ok) {
#pragma unroll
                    for (int t = 0; t < TT; ++t)
                        if (t < nvalid) drow[(unsigned)(tl + t)] = dv[t];      // unsigned: scalar base + 32-bit lane offset
                }
            }
            if (!last_pass) {
                park[(5 * k + 0) * blockDim.x] = rb; park[(5 * k + 1) * blockDim.x] = rd;
                park[(5 * k + 2) * blockDim.x] = rZ; park[(5 * k + 3) * blockDim.x] = rM;
                park[(5 * k + 4) * blockDim.x] = __int_as_float(ri);
            } else {
                // ---- merge the 64 lanes of the row (results land in lane 63)
                float best = rb;
                int idx = ri;
                IGN_ARGMAX_STEP(DPP_QP_XOR1, 0xf);
                IGN_ARGMAX_STEP(DPP_QP_XOR2, 0xf);
                IGN_ARGMAX_STEP(DPP_HALF_MIRROR, 0xf);
                IGN_ARGMAX_STEP(DPP_MIRROR, 0xf);
                IGN_ARGMAX_STEP(DPP_BCAST15, 0xa);
                IGN_ARGMAX_STEP(DPP_BCAST31, 0xc);
                float dmin = wave_min_l63(rd);
                float Z = rZ, M = rM;
                if (ae->gate == GATE_LTS) {
                    dmin = wave_bcast_l63(dmin);                  // every lane rescales its partial sums to the row minimum
                    const float sc = (rd < INFINITY) ? __expf(dmin - rd) : 0.f;
                    Z *= sc; M *= sc;
                }
                Z = wave_sum_l63(Z);
                M = wave_sum_l63(M);
                if (lane == 63 && row_ok) {
                    const int kk = k0 + k;
                    const size_t col = (size_t)be * ae->ld + ae->col0 + (size_t)kk * ae->C + c;
                    const size_t sidx = ((size_t)be * ae->K + kk) * ae->C + c;
                    float pout;
                    if (ae->gate == GATE_RBF) {
                        pout = best;                       // = p[t*] * (1 + s - s): Shapelet.py:81-82
                    } else {
                        const float th = ae->thr[(size_t)kk * ae->C + c];
                        pout = 1.f / (1.f + __expf(-(th - dmin)));
                    }
                    ae->p_out[col] = pout;
                    ae->dmin_out[col] = dmin;
                    ae->tstar[sidx] = idx;
                    ae->zmu[2 * sidx] = Z;
                    ae->zmu[2 * sidx + 1] = M / Z;
                }
            }
        }
    }
}

template <int TT, int KT, int DIST>
__global__ void __launch_bounds__(256, 4) shp_fwd_kernel(const ShpFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    shp_fwd_body<TT, KT, DIST>(a, blockIdx.x, blockIdx.y, smem);
}

template <int TT, int KT, int DIST>
static void shp_fwd_launch(const ShpFwdArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024)      // long rows (EigenWorms: T = 17 984 = 72 KB): a CU of gfx950 has 160 KB of LDS, the default cap is 64
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&shp_fwd_kernel<TT, KT, DIST>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((shp_fwd_kernel<TT, KT, DIST>), grid, block, lds, s, a);
}

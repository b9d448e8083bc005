// Shapelet backward (dloss/dw): second sliding pass with the roles of t and j swapped.
//
// Autograd of IGN/model/Shapelet.py:60-84 / :96-111, closed form in SURVEY.md App. A:
//   dl/dw[k,c,j] = sum_b sum_t  A[b,k,c,t] * sign(x[b,c,t+j] - w[k,c,j])           (L1)
//   dl/dw[k,c,j] = sum_b sum_t A2[b,k,c,t] * (x[b,c,t+j] - w[k,c,j])                (MSE)
// where A = -(dl/dd_t)/L and dl/dd_t follows from the saved distances d_t, the row statistics
// {t*, Z, mu} and the upstream gradient of the gate output.
//
// Mapping (fp32 VALU bound).  L1 inner step per (b,c,k,t,j) element: `v_cmpx_gt_f32 x, w` (EXEC <- x > w),
// `v_add_f32 acc, acc, A` under that mask, `s_mov_b64 exec, -1` -- 2 VALU + 1 SALU instead of the compiler's
// v_cmp / s_nop / v_cndmask / v_add (3 VALU + a hazard nop); measured 24 vs 19-20 T elements/s
// (profiles/r1_valu_microbench.txt).  It accumulates P_j = sum_{t: x>w} A_t; with S = sum_t A_t the signed sum is
// sum_t A_t sign(x-w) = 2 P_j - S.
//   * a block owns one channel c (and a tile of `kb` shapelets) and walks a slice of the batch;
//   * lane <-> (k, chunk of JJ consecutive j): its JJ accumulators and JJ weights stay in registers for
//     the whole batch slice, so there is NO cross-lane reduction -- only a fixed-order reduction over
//     batch slices afterwards (deterministic, no float atomics);
//   * per batch row and per chunk of `tc` window positions the block stages x[b,c,t0:t0+tc+L) and
//     A[k,t0:t0+tc) (computed cooperatively from d, once per (b,k,c,t), with its per-shapelet sum S) in LDS; each
//     lane then slides over the chunk in steps of JJ with a ping-pong register window of 2*JJ samples (no register
//     moves), reading A[k][t..t+JJ) as an LDS broadcast;
//   * the v_cmpx sequence is latency-bound per wave (EXEC round trip), so the kernel is built for occupancy:
//     <= 72 VGPRs and an LDS budget of ~5 KB per wave (the t-chunking) give 7-8 waves per SIMD.  Blocks are kept as
//     small as lane utilisation allows (one wave per shapelet at L=500: 17 -> 24 T elements/s vs a 5-wave block).
//     Measured alternatives (tests/diag_shapelet.py, ms per step at the benchmark shape): 16-step unchunked 6.9;
//     chunked multi-wave 6.3; + small blocks 5.66 (this); one-wave blocks straddling shapelets 6.2 (A staged twice).
#pragma once
#include "ign_common.h"

template <int JJ>
__device__ __forceinline__ void bwd_l1_step(float (&acc)[JJ], const float (&wa)[JJ], const float (&wb)[JJ],
                                            const float (&wreg)[JJ], const float (&A)[JJ]) {
    // window sample for (t, jj) is wa[t + jj] when t + jj < JJ, else wb[t + jj - JJ]
#pragma unroll
    for (int t = 0; t < JJ; ++t) {
#pragma unroll
        for (int jj = 0; jj < JJ; jj += 4) {
#define IGN_WIN(q) ((t + jj + (q)) < JJ ? wa[(t + jj + (q)) % JJ] : wb[(t + jj + (q)) % JJ])
            // one asm statement per 4 elements: hipcc pads every asm boundary with an s_nop
            asm volatile(
                "v_cmpx_gt_f32 %4, %8\n\tv_add_f32 %0, %0, %12\n\ts_mov_b64 exec, -1\n\t"
                "v_cmpx_gt_f32 %5, %9\n\tv_add_f32 %1, %1, %12\n\ts_mov_b64 exec, -1\n\t"
                "v_cmpx_gt_f32 %6, %10\n\tv_add_f32 %2, %2, %12\n\ts_mov_b64 exec, -1\n\t"
                "v_cmpx_gt_f32 %7, %11\n\tv_add_f32 %3, %3, %12\n\ts_mov_b64 exec, -1"
                : "+v"(acc[jj]), "+v"(acc[jj + 1]), "+v"(acc[jj + 2]), "+v"(acc[jj + 3])
                : "v"(IGN_WIN(0)), "v"(IGN_WIN(1)), "v"(IGN_WIN(2)), "v"(IGN_WIN(3)),
                  "v"(wreg[jj]), "v"(wreg[jj + 1]), "v"(wreg[jj + 2]), "v"(wreg[jj + 3]), "v"(A[t])
                : "vcc");
#undef IGN_WIN
        }
    }
}

template <int JJ>
__device__ __forceinline__ void bwd_mse_step(float (&acc)[JJ], const float (&wa)[JJ], const float (&wb)[JJ],
                                             const float (&wreg)[JJ], const float (&A)[JJ]) {
#pragma unroll
    for (int t = 0; t < JJ; ++t)
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            const float xv = (t + jj) < JJ ? wa[(t + jj) % JJ] : wb[(t + jj) % JJ];
            acc[jj] = fmaf(A[t], xv - wreg[jj], acc[jj]);
        }
}

template <int JJ>
__device__ __forceinline__ void bwd_dot_step(float (&acc)[JJ], const float (&wa)[JJ], const float (&wb)[JJ],
                                             const float (&A)[JJ]) {
#pragma unroll
    for (int t = 0; t < JJ; ++t)
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) {
            const float xv = (t + jj) < JJ ? wa[(t + jj) % JJ] : wb[(t + jj) % JJ];
            acc[jj] = fmaf(A[t], xv, acc[jj]);
        }
}

template <int JJ>
__device__ __forceinline__ void lds_load(float (&dst)[JJ], const float* p) {
#pragma unroll
    for (int i = 0; i < JJ; i += 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        dst[i] = v.x; dst[i + 1] = v.y; dst[i + 2] = v.z; dst[i + 3] = v.w;
    }
}

// STRIDED (window step > 1, the reference's rule once seq_len >= 3000): consecutive windows no longer share samples, so
// the ping-pong register window is replaced by a plain loop -- per window position t the lane reads its JJ samples
// x[t*stride + j] from the LDS chunk and A[k][t] as a broadcast.  A shapelet longer than 512*JJ positions is split over
// `njt` blocks (blockIdx.z = k*njt + tile), each recomputing A for its shapelet (kb = 1).  Same staging, same fixed-order
// reduction, same outputs; built for the long-sequence UEA sets (MotorImagery, EigenWorms), not for the benchmark shape.
template <int JJ, int DIST, bool STRIDED = false>
__global__ void __launch_bounds__(512, 7) shp_bwd_kernel(const ShpBwdArgs a) {
    static_assert(JJ % 4 == 0, "float4 LDS reads need 4-float alignment");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                        // [xs_len]      x[b,c,t0 + i]
    float* As = smem + a.xs_len;             // [kb][tc]      A[k][t0 + t]
    float* Sp = As + a.kb * a.tc;            // [kb][8]       per-wave partial sums of A (fixed order -> deterministic)
    float* Par = Sp + a.kb * 8;              // [kb][8]       per-shapelet scalars of the current row (g, t*, 1/Z, mu, ...)

    const int c = blockIdx.x, bs = blockIdx.y;
    const int ztile = STRIDED ? (int)blockIdx.z / a.njt : (int)blockIdx.z;
    const int j0 = STRIDED ? ((int)blockIdx.z - ztile * a.njt) * a.cpk * JJ : 0;      // first shapelet position of this tile
    const int kbase = ztile * a.kb;
    const int kcount = min(a.kb, a.K - kbase);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    int kl = tid / a.cpk;
    const int jc = tid - kl * a.cpk;
    const bool active = kl < kcount;
    if (!active) kl = kcount - 1;            // padding lanes shadow a valid shapelet: EXEC stays full in the hot loop
    const int jbase = jc * JJ;                 // position inside the tile; the shapelet position is j0 + jbase + jj

    float wreg[JJ], acc[JJ];
    float ssum = 0.f;                        // S = sum_t A_t of this lane's shapelet over the batch slice (L1)
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) {
        const int j = j0 + jbase + jj;
        wreg[jj] = (j < a.L) ? a.w[((size_t)(kbase + kl) * a.C + c) * a.L + j] : INFINITY;
        acc[jj] = 0.f;
    }

    const int b_begin = (int)(((long long)a.B * bs) / a.nbs);
    const int b_end = (int)(((long long)a.B * (bs + 1)) / a.nbs);
    const float two_eps2 = 2.f * a.eps * a.eps;

    for (int b = b_begin; b < b_end; ++b) {
        const float* row = a.xn + ((size_t)b * a.C + c) * a.T;
        for (int t0 = 0; t0 < a.Tw; t0 += a.tc) {
            __syncthreads();                 // previous chunk fully consumed
            // ---- staging, arranged so that global-memory latency is paid a few times per chunk, not once per element:
            // s_memtime stamps showed 25k of a chunk's 60-135k cycles in this phase, almost all of it ~20 serialised
            // load round trips (per shapelet: its scalars, then load -> exp -> store per window position).  Now the x
            // chunk and the per-shapelet scalars are fetched together (one round trip), the scalars go through LDS, and
            // the saved distances are loaded in batches of SB positions for SG shapelets before any of them is used.
            constexpr int XB = 4;
            for (int i0 = tid; i0 < a.xs_len; i0 += XB * nthr) {
                float xv[XB];
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int i = i0 + u * nthr;
                    const int src = (STRIDED ? t0 * a.stride + j0 : t0) + i;
                    xv[u] = (i < a.xs_len && src < a.T) ? row[src] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < XB; ++u) {
                    const int i = i0 + u * nthr;
                    if (i < a.xs_len) xs[i] = xv[u];
                }
            }
            if (tid < kcount) {
                const int k = kbase + tid;
                const size_t sidx = ((size_t)b * a.K + k) * a.C + c;
                const size_t col = (size_t)b * a.ld + a.col0 + (size_t)k * a.C + c;
                const float gv = a.g[col];
                const int ts = a.tstar[sidx];
                const float z = a.zmu[2 * sidx];
                const float mu = a.zmu[2 * sidx + 1];
                float gm = 0.f, dmin = 0.f;
                if (a.gate == GATE_LTS) {
                    const float P = a.p[col];
                    gm = -gv * P * (1.f - P);       // dP/dm = -sigma'(thr - m)
                    dmin = a.dmin[col];
                }
                float* pr = Par + tid * 8;
                pr[0] = gv; pr[1] = __int_as_float(ts); pr[2] = 1.f / z; pr[3] = mu; pr[4] = gm; pr[5] = dmin;
                pr[6] = (DIST >= DIST_COS) ? a.wnorm[(size_t)k * a.C + c] : 0.f;
            }
            __syncthreads();
            {
                constexpr int SG = 5, SB = 2;            // shapelets x window positions fetched together (10 loads in flight)
                const float* xst = (DIST >= DIST_COS) ? a.xstat + ((size_t)b * a.C + c) * a.Tw : nullptr;
                const float* dbase = a.d + (((size_t)b * a.C + c) * a.K + kbase) * a.Tw;
                for (int kg = 0; kg < kcount; kg += SG) {
                    float part[SG];
#pragma unroll
                    for (int q = 0; q < SG; ++q) part[q] = 0.f;
                    for (int tb = tid; tb < a.tc; tb += SB * nthr) {
                        float dv[SG][SB], xsv[SB];
#pragma unroll
                        for (int q = 0; q < SG; ++q)
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const int tt = tb + u * nthr, t = t0 + tt;
                                dv[q][u] = (kg + q < kcount && tt < a.tc && t < a.Tw) ? dbase[(size_t)(kg + q) * a.Tw + t] : 0.f;
                            }
                        if (DIST >= DIST_COS) {
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const int tt = tb + u * nthr, t = t0 + tt;
                                xsv[u] = (tt < a.tc && t < a.Tw) ? xst[t] : 1.f;
                            }
                        }
#pragma unroll
                        for (int q = 0; q < SG; ++q) {
                            if (kg + q >= kcount) break;
                            const float* pr = Par + (kg + q) * 8;
                            const float gv = pr[0], invZ = pr[2], mu = pr[3], gm = pr[4], dmin = pr[5], wn = pr[6];
                            const int ts = __float_as_int(pr[1]);
                            float* Ak = As + (kg + q) * a.tc;
#pragma unroll
                            for (int u = 0; u < SB; ++u) {
                                const int tt = tb + u * nthr, t = t0 + tt;
                                if (tt >= a.tc) break;
                                float A = 0.f;
                                if (t < a.Tw) {
                                    const float d1 = dv[q][u];
                                    float dldd;
                                    if (a.gate == GATE_RBF) {
                                        const float uu = a.eps * d1;
                                        const float pp = __expf(-(uu * uu));
                                        const float e = __expf(pp);
                                        const float coef = gv * ((t == ts ? 1.f : 0.f) + e * invZ * (pp - mu));
                                        dldd = coef * (-two_eps2 * d1 * pp);
                                    } else {
                                        const float sft = __expf(dmin - d1) * invZ;
                                        dldd = gm * ((t == ts ? 1.f : 0.f) + sft * (mu - d1));
                                    }
                                    if (DIST == DIST_L1) {
                                        A = -dldd * a.invL;
                                        part[q] += A;
                                    } else if (DIST == DIST_MSE) {
                                        A = -2.f * dldd * a.invL;
                                    } else if (DIST == DIST_COS) {
                                        // d = 1 - <x,w>/(max(|x|,e) max(|w|,e)):  dd/dw_j = -x_j/den + cos * w_j / |w|^2
                                        A = -dldd / (fmaxf(xsv[u], 1e-8f) * fmaxf(wn, 1e-8f));
                                        if (wn >= 1e-8f) part[q] += dldd * (1.f - d1) / (wn * wn);
                                    } else {
                                        // d = 1 - <x,wc>/(sx sw + e):  dd/dwc_j = -x_j/den + rho * sx * wc_j / (sw den)
                                        const float den = xsv[u] * wn + 1e-8f;
                                        A = -dldd / den;
                                        if (wn > 0.f) part[q] += dldd * (1.f - d1) * xsv[u] / (wn * den);
                                    }
                                }
                                Ak[tt] = A;
                                __builtin_amdgcn_sched_barrier(0);   // one element at a time: keeps the register budget of the main loop
                            }
                        }
                    }
                    if (DIST != DIST_MSE) {
#pragma unroll
                        for (int q = 0; q < SG; ++q) {
                            if (kg + q >= kcount) break;
                            float pq = part[q];
#pragma unroll
                            for (int o = 32; o > 0; o >>= 1) pq += __shfl_xor(pq, o, 64);
                            if (lane == 0) Sp[(kg + q) * 8 + wave] = pq;
                        }
                    }
                }
            }
            __syncthreads();

            if (DIST != DIST_MSE)
                for (int wv = 0; wv < nwave; ++wv) ssum += Sp[kl * 8 + wv];
            const float* Ak = As + kl * a.tc;
            const float* xl = xs + jbase;
            if (STRIDED) {
                for (int t = 0; t < a.tc; ++t) {             // A is 0 beyond Tw, x is 0 beyond T
                    const float At = Ak[t];
                    const float* xw = xl + t * a.stride;
#pragma unroll
                    for (int jj = 0; jj < JJ; ++jj) {
                        const float xv = xw[jj];
                        if (DIST == DIST_L1)       acc[jj] += (xv > wreg[jj]) ? At : 0.f;
                        else if (DIST == DIST_MSE) acc[jj] = fmaf(At, xv - wreg[jj], acc[jj]);
                        else                       acc[jj] = fmaf(At, xv, acc[jj]);
                    }
                }
                continue;
            }
            float Wa[JJ], Wb[JJ], A[JJ];
            lds_load<JJ>(Wa, xl);
            for (int t = 0; t < a.tc; t += 2 * JJ) {         // tc is a multiple of 2*JJ: two ping-pong steps
                lds_load<JJ>(Wb, xl + t + JJ);
                lds_load<JJ>(A, Ak + t);
                if (DIST == DIST_L1)       bwd_l1_step<JJ>(acc, Wa, Wb, wreg, A);
                else if (DIST == DIST_MSE) bwd_mse_step<JJ>(acc, Wa, Wb, wreg, A);
                else                       bwd_dot_step<JJ>(acc, Wa, Wb, A);
                lds_load<JJ>(Wa, xl + t + 2 * JJ);
                lds_load<JJ>(A, Ak + t + JJ);
                if (DIST == DIST_L1)       bwd_l1_step<JJ>(acc, Wb, Wa, wreg, A);
                else if (DIST == DIST_MSE) bwd_mse_step<JJ>(acc, Wb, Wa, wreg, A);
                else                       bwd_dot_step<JJ>(acc, Wb, Wa, A);
            }
        }
    }

    if (active) {
        float* out = a.part + (((size_t)bs * a.K + (kbase + kl)) * a.C + c) * a.L;
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj)
            if (j0 + jbase + jj < a.L)
                out[j0 + jbase + jj] = (DIST == DIST_L1) ? 2.f * acc[jj] - ssum
                                : (DIST == DIST_MSE) ? acc[jj] : acc[jj] + wreg[jj] * ssum;
    }
}

template <int JJ, int DIST>
static void shp_bwd_launch(const ShpBwdArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((shp_bwd_kernel<JJ, DIST, false>), grid, block, lds, s, a);
}

template <int DIST>
static void shp_bwd_strided_launch(const ShpBwdArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((shp_bwd_kernel<4, DIST, true>), grid, block, lds, s, a);
}

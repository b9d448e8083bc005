// Expert-head GEMMs, the gini gate and the flat Adam update: the small HBM-bound pieces of the training step.
//
//  * head GEMM  out[b,n] = sum_f X[b,f] W[n,f] (+ bias[n])  with N = number of classes (3..16) -- "skinny":
//    IGN/model/Shapelet.py:171,200 (SBM head, F = G*K*C = 2440), IGN/model/Transformer.py:72,109 (F = T*d = 512000),
//    IGN/model/FullyConvNet.py:50,58.  N is far too small for an MFMA tile to pay (a 32x32 tile would be >90 % padding)
//    and the op moves 4*(B*F + N*F) bytes for 2*B*F*N flops (intensity ~N/2 flop/byte): HBM/L2 bound, so it is a
//    coalesced float4 streaming kernel with N accumulators per thread and a block reduction.
//  * gini gate  eta = (N*sum softmax(s)^2 - 1)/(N-1); out = eta*s + (1-eta)*d   IGN/model/InterpGN.py:44-52, fwd + bwd.
//  * Adam       one launch over the flat parameter / gradient / moment buffers (torch.optim.Adam semantics,
//    IGN/exp/experiment_classification.py:136,338).
#include "ign_common.h"

constexpr int HEAD_NMAX = 16;

// ------------------------------------------------------------------------------------------------ head forward
// One block per row; 256 threads, or 1024 for long rows (the Transformer's 512 000-feature head: one 256-thread block per CU
// kept four loads per lane in flight and ran at 0.7 TB/s).
__global__ void __launch_bounds__(1024) head_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int B, int F, int N, long long ldx) {
    __shared__ float red[16][HEAD_NMAX];
    const int b = blockIdx.x;
    const float* x = X + (long long)b * ldx;
    float acc[HEAD_NMAX];
#pragma unroll
    for (int n = 0; n < HEAD_NMAX; ++n) acc[n] = 0.f;
    const int F4 = F & ~3;
    const int nthr = blockDim.x;
    for (int f = threadIdx.x * 4; f < F4; f += nthr * 4) {
        const float4 xv = *reinterpret_cast<const float4*>(x + f);
#pragma unroll
        for (int n = 0; n < HEAD_NMAX; ++n)
            if (n < N) {
                const float4 wv = *reinterpret_cast<const float4*>(W + (long long)n * F + f);
                acc[n] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
            }
    }
    for (int f = F4 + threadIdx.x; f < F; f += nthr)
#pragma unroll
        for (int n = 0; n < HEAD_NMAX; ++n)
            if (n < N) acc[n] += x[f] * W[(long long)n * F + f];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int n = 0; n < HEAD_NMAX; ++n) {
        float v = acc[n];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][n] = v;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        const int n = threadIdx.x;
        float t = 0.f;
        for (int w = 0; w < (nthr >> 6); ++w) t += red[w][n];          // fixed order
        out[(long long)b * N + n] = t + (bias ? bias[n] : 0.f);
    }
}

// gX[b,f] = sum_n g[b,n] W[n,f]
__device__ __forceinline__ void head_bwd_x_body(const float* __restrict__ g, const float* __restrict__ W,
                                                float* __restrict__ gX, int B, int F, int N, long long ldx, int bx, int b) {
    const int f = (bx * 256 + threadIdx.x) * 4;
    if (f >= F) return;
    float gn[HEAD_NMAX];
#pragma unroll
    for (int n = 0; n < HEAD_NMAX; ++n) gn[n] = n < N ? g[(long long)b * N + n] : 0.f;
    if (f + 3 < F) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int n = 0; n < HEAD_NMAX; ++n)
            if (n < N) {
                const float4 wv = *reinterpret_cast<const float4*>(W + (long long)n * F + f);
                r.x += gn[n] * wv.x; r.y += gn[n] * wv.y; r.z += gn[n] * wv.z; r.w += gn[n] * wv.w;
            }
        *reinterpret_cast<float4*>(gX + (long long)b * ldx + f) = r;
    } else {
        for (int ff = f; ff < F; ++ff) {
            float r = 0.f;
            for (int n = 0; n < N; ++n) r += gn[n] * W[(long long)n * F + ff];
            gX[(long long)b * ldx + ff] = r;
        }
    }
}

__global__ void __launch_bounds__(256) head_bwd_x_kernel(const float* __restrict__ g, const float* __restrict__ W,
                                                         float* __restrict__ gX, int B, int F, int N, long long ldx) {
    head_bwd_x_body(g, W, gX, B, F, N, ldx, blockIdx.x, blockIdx.y);
}

// gW[n,f] = sum_b g[b,n] X[b,f];  gbias[n] = sum_b g[b,n].
// Block = 32 consecutive f x 8 batch groups: each thread sums its group's rows in ascending order, the 8 partials are
// combined in fixed order through LDS (deterministic).  32 f per block keeps F/32 blocks in flight (F=2440: 77 blocks;
// a 256-f-per-block version ran 10 blocks and took 125 us of a 17.9 ms step).
// `add` / `add_scale` (both nullable): gW += add_scale[0] * add -- a gradient of the same tensor that does not depend on the batch
// (the L1 regulariser of the SBM head, IGN/model/Shapelet.py:219) rides on this store instead of an accumulate kernel.
__device__ __forceinline__ void head_bwd_w_body(const float* __restrict__ g, const float* __restrict__ X,
                                                float* __restrict__ gW, float* __restrict__ gbias, int B, int F,
                                                int N, long long ldx, const float* __restrict__ add,
                                                const float* __restrict__ add_scale, const int bx) {
    extern __shared__ float sm[];               // g copy [B*N], then partials [8][HEAD_NMAX][32]
    float* gs = sm;
    float* part = sm + B * N;
    for (int i = threadIdx.x; i < B * N; i += 256) gs[i] = g[i];
    __syncthreads();
    if (gbias && bx == 0 && threadIdx.x < N) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += gs[b * N + threadIdx.x];
        gbias[threadIdx.x] = s;
    }
    const int fl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int f = bx * 32 + fl;
    const int per = (B + 7) / 8;
    const int b0 = grp * per, b1 = min(B, b0 + per);
    float acc[HEAD_NMAX];
#pragma unroll
    for (int n = 0; n < HEAD_NMAX; ++n) acc[n] = 0.f;
    if (f < F)
        for (int b = b0; b < b1; ++b) {
            const float xv = X[(long long)b * ldx + f];
#pragma unroll
            for (int n = 0; n < HEAD_NMAX; ++n)
                if (n < N) acc[n] = fmaf(gs[b * N + n], xv, acc[n]);
        }
#pragma unroll
    for (int n = 0; n < HEAD_NMAX; ++n)
        if (n < N) part[(grp * HEAD_NMAX + n) * 32 + fl] = acc[n];
    __syncthreads();
    for (int i = threadIdx.x; i < N * 32; i += 256) {
        const int n = i >> 5, ff = i & 31;
        if (bx * 32 + ff < F) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += part[(q * HEAD_NMAX + n) * 32 + ff];
            const long long o = (long long)n * F + bx * 32 + ff;
            if (add) s += (add_scale ? add_scale[0] : 1.f) * add[o];
            gW[o] = s;
        }
    }
}

__global__ void __launch_bounds__(256) head_bwd_w_kernel(const float* __restrict__ g, const float* __restrict__ X,
                                                         float* __restrict__ gW, float* __restrict__ gbias, int B, int F,
                                                         int N, long long ldx, const float* __restrict__ add,
                                                         const float* __restrict__ add_scale) {
    head_bwd_w_body(g, X, gW, gbias, B, F, N, ldx, add, add_scale, blockIdx.x);
}

// both gradients in ONE launch: blocks [0, nwb) play head_bwd_w_kernel's role, the rest head_bwd_x_kernel's ((nxb, B) grid
// flattened) -- block-uniform branch, the two halves touch disjoint outputs
__global__ void __launch_bounds__(256) head_bwd_xw_kernel(const float* __restrict__ g, const float* __restrict__ X,
                                                          const float* __restrict__ W, float* __restrict__ gX,
                                                          float* __restrict__ gW, float* __restrict__ gbias, int B, int F, int N,
                                                          long long ldx, const float* __restrict__ add,
                                                          const float* __restrict__ add_scale, int nwb, int nxb) {
    const int blk = blockIdx.x;
    if (blk < nwb) {
        head_bwd_w_body(g, X, gW, gbias, B, F, N, ldx, add, add_scale, blk);
    } else {
        const int r = blk - nwb;
        head_bwd_x_body(g, W, gX, B, F, N, ldx, r % nxb, r / nxb);
    }
}

// ------------------------------------------------------------------------------------------------ gini gate
// forward: q = softmax(s); G = sum q^2; eta = (N G - 1)/(N - 1); [eta > thr -> 1]; out = eta s + (1 - eta) d
// backward: d eta / d s_j = (2N/(N-1)) q_j (q_j - G); ds = eta*gout + (sum_n gout_n (s_n - d_n)) * deta/ds ; dd = (1-eta)*gout
__global__ void __launch_bounds__(256) gate_fwd_kernel(const float* __restrict__ s, const float* __restrict__ d,
                                                       float* __restrict__ out, float* __restrict__ eta_out, int B, int N,
                                                       float thr, int use_thr) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* sr = s + (long long)b * N;
    float mx = -INFINITY;
    for (int n = 0; n < N; ++n) mx = fmaxf(mx, sr[n]);
    float z = 0.f, z2 = 0.f;
    for (int n = 0; n < N; ++n) {
        const float e = expf(sr[n] - mx);
        z += e;
        z2 += e * e;
    }
    float eta = ((float)N * (z2 / (z * z)) - 1.f) / (float)(N - 1);
    if (use_thr && eta > thr) eta = 1.f;
    eta_out[b] = eta;
    for (int n = 0; n < N; ++n) out[(long long)b * N + n] = eta * sr[n] + (1.f - eta) * d[(long long)b * N + n];
}

__global__ void __launch_bounds__(256) gate_bwd_kernel(const float* __restrict__ s, const float* __restrict__ d,
                                                       const float* __restrict__ gout, const float* __restrict__ geta,
                                                       float* __restrict__ gs, float* __restrict__ gd, int B, int N,
                                                       float thr, int use_thr) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* sr = s + (long long)b * N;
    const float* dr = d + (long long)b * N;
    const float* gr = gout + (long long)b * N;
    float mx = -INFINITY;
    for (int n = 0; n < N; ++n) mx = fmaxf(mx, sr[n]);
    float z = 0.f, z2 = 0.f, dot = geta ? geta[b] : 0.f;
    for (int n = 0; n < N; ++n) {
        const float e = expf(sr[n] - mx);
        z += e;
        z2 += e * e;
        dot += gr[n] * (sr[n] - dr[n]);
    }
    const float G = z2 / (z * z);
    float eta = ((float)N * G - 1.f) / (float)(N - 1);
    float c = 2.f * (float)N / (float)(N - 1) * dot;
    if (use_thr && eta > thr) { eta = 1.f; c = 0.f; }         // the hard branch has no gradient through eta
    for (int n = 0; n < N; ++n) {
        const float q = expf(sr[n] - mx) / z;
        gs[(long long)b * N + n] = eta * gr[n] + c * q * (q - G);
        gd[(long long)b * N + n] = (1.f - eta) * gr[n];
    }
}

// ------------------------------------------------------------------------------------------------ fused training-loss tail
// IGN's per-step loss tail, IGN/exp/experiment_classification.py:320-329 + IGN/model/InterpGN.py:44-52, in ONE launch:
//   out = eta*s + (1-eta)*d (gini gate);  loss = CE(out, y) + beta*CE(s, y)  (batch means);  and the gradients of that
//   loss w.r.t. both experts' logits -- what torch spends ~40 softmax / nll / mean / add kernels (forward + backward) on,
//   all of them serialised between the last forward kernel and the first backward kernel.
// One block; thread <-> rows b, b+256, ...; the two CE sums are reduced through LDS in thread order (deterministic).
constexpr int LOSS_NMAX = 16;
__global__ void __launch_bounds__(256) ign_loss_kernel(const float* __restrict__ s, const float* __restrict__ d,
                                                       const long long* __restrict__ y, float* __restrict__ out,
                                                       float* __restrict__ eta_out, float* __restrict__ loss2,
                                                       float* __restrict__ gs, float* __restrict__ gd, int B, int N, float beta,
                                                       const float* __restrict__ reg) {
    __shared__ float red[2][256];
    float ce_o = 0.f, ce_s = 0.f;
    const float invB = 1.f / (float)B;
    for (int b = threadIdx.x; b < B; b += 256) {
        float sv[LOSS_NMAX], ov[LOSS_NMAX], q[LOSS_NMAX];
        const float* sr = s + (long long)b * N;
        const float* dr = d + (long long)b * N;
        const int yb = (int)y[b];
        float mx = -INFINITY;
        for (int n = 0; n < N; ++n) { sv[n] = sr[n]; mx = fmaxf(mx, sv[n]); }
        float z = 0.f, z2 = 0.f;
        for (int n = 0; n < N; ++n) { q[n] = expf(sv[n] - mx); z += q[n]; z2 += q[n] * q[n]; }
        const float G = z2 / (z * z);
        const float eta = ((float)N * G - 1.f) / (float)(N - 1);
        eta_out[b] = eta;
        const float lse_s = mx + logf(z);
        ce_s += lse_s - sv[yb];
        float mo = -INFINITY;
        for (int n = 0; n < N; ++n) {
            ov[n] = eta * sv[n] + (1.f - eta) * dr[n];
            out[(long long)b * N + n] = ov[n];
            mo = fmaxf(mo, ov[n]);
        }
        float zo = 0.f;
        for (int n = 0; n < N; ++n) zo += expf(ov[n] - mo);
        ce_o += mo + logf(zo) - ov[yb];
        // gradients: g_out = (softmax(out) - onehot)/B ; through the gate (see gate_bwd_kernel) ; + beta*(softmax(s) - onehot)/B
        float dot = 0.f;
        float go[LOSS_NMAX];
        for (int n = 0; n < N; ++n) {
            go[n] = (expf(ov[n] - mo) / zo - (n == yb ? 1.f : 0.f)) * invB;
            dot += go[n] * (sv[n] - dr[n]);
        }
        const float c = 2.f * (float)N / (float)(N - 1) * dot;
        for (int n = 0; n < N; ++n) {
            const float qn = q[n] / z;
            gs[(long long)b * N + n] = eta * go[n] + c * qn * (qn - G) + beta * (qn - (n == yb ? 1.f : 0.f)) * invB;
            gd[(long long)b * N + n] = (1.f - eta) * go[n];
        }
    }
    red[0][threadIdx.x] = ce_o;
    red[1][threadIdx.x] = ce_s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b2 = 0.f;
        for (int i = 0; i < 256; ++i) { a += red[0][i]; b2 += red[1][i]; }
        loss2[0] = a * invB;
        loss2[1] = b2 * invB;
        loss2[2] = a * invB + beta * (b2 * invB) + (reg ? reg[0] : 0.f);        // + info.loss.mean() (exp:325-329)
    }
}

extern "C" int ign_loss_fwd_bwd_reg(const float* sbm, const float* dnn, const long long* labels, const float* reg, float* out,
                                    float* eta, float* loss2, float* gsbm, float* gdnn, int B, int N, float beta, void* stream) {
    if (!sbm || !dnn || !labels || !out || !eta || !loss2 || !gsbm || !gdnn || B <= 0 || N < 2 || N > LOSS_NMAX) {
        ign_set_error("ign_loss_fwd_bwd: null pointer or bad dimension (B=%d N=%d, N <= %d)", B, N, LOSS_NMAX);
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(ign_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sbm, dnn, labels, out, eta, loss2, gsbm, gdnn, B, N,
                       beta, reg);
    return ign_check_launch("ign_loss_kernel");
}

extern "C" int ign_loss_fwd_bwd(const float* sbm, const float* dnn, const long long* labels, float* out, float* eta, float* loss2,
                                float* gsbm, float* gdnn, int B, int N, float beta, void* stream) {
    return ign_loss_fwd_bwd_reg(sbm, dnn, labels, nullptr, out, eta, loss2, gsbm, gdnn, B, N, beta, stream);
}

// ------------------------------------------------------------------------------------------------ Adam
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2,
                                                   float eps, float bc1, float bc2_sqrt) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float step = lr / bc1;
    if (i + 3 < n) {
        float4 pv = *reinterpret_cast<float4*>(p + i);
        const float4 gv = *reinterpret_cast<const float4*>(g + i);
        float4 mv = *reinterpret_cast<float4*>(m + i);
        float4 vv = *reinterpret_cast<float4*>(v + i);
#define ADAM1(P, G, M, V)                                   \
        M = b1 * M + (1.f - b1) * G;                         \
        V = b2 * V + (1.f - b2) * G * G;                     \
        P -= step * M / (sqrtf(V) / bc2_sqrt + eps);
        ADAM1(pv.x, gv.x, mv.x, vv.x) ADAM1(pv.y, gv.y, mv.y, vv.y) ADAM1(pv.z, gv.z, mv.z, vv.z) ADAM1(pv.w, gv.w, mv.w, vv.w)
        *reinterpret_cast<float4*>(p + i) = pv;
        *reinterpret_cast<float4*>(m + i) = mv;
        *reinterpret_cast<float4*>(v + i) = vv;
    } else {
        for (long long j = i; j < n; ++j) {
            float pv = p[j], gv = g[j], mv = m[j], vv = v[j];
            ADAM1(pv, gv, mv, vv)
            p[j] = pv; m[j] = mv; v[j] = vv;
        }
    }
}

// Gather per-parameter gradient tensors into the flat bucket in ONE launch (instead of one accumulate kernel per parameter):
// blockIdx.y = table entry, blockIdx.x strides over its elements.
constexpr int GATHER_MAX = 96;
struct GatherTable {
    const float* src[GATHER_MAX];
    long long off[GATHER_MAX];
    long long n[GATHER_MAX];
};
__global__ void __launch_bounds__(256) gather_flat_kernel(const GatherTable t, float* __restrict__ flat) {
    const int e = blockIdx.y;
    const float* __restrict__ src = t.src[e];
    float* __restrict__ dst = flat + t.off[e];
    const long long n = t.n[e];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

// Graph-capturable Adam: the step count lives on the device, so a captured launch sequence stays valid when replayed.
__global__ void adam_tick_kernel(int* __restrict__ step_dev, float* __restrict__ bc_dev, float b1, float b2) {
    const int step = ++(*step_dev);
    bc_dev[0] = (float)(1.0 - pow((double)b1, (double)step));
    bc_dev[1] = (float)sqrt(1.0 - pow((double)b2, (double)step));
}

__global__ void __launch_bounds__(256) adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long long n, float lr, float b1, float b2,
                                                       float eps, const float* __restrict__ bc_dev) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float bc1 = bc_dev[0], bc2_sqrt = bc_dev[1];
    const float step = lr / bc1;
    for (long long j = i; j < min(n, i + 4); ++j) {
        float pv = p[j], gv = g[j], mv = m[j], vv = v[j];
        ADAM1(pv, gv, mv, vv)
        p[j] = pv; m[j] = mv; v[j] = vv;
    }
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int ign_head_fwd(const float* X, const float* W, const float* bias, float* out, int B, int F, int N,
                            long long ldx, void* stream) {
    if (!X || !W || !out || B <= 0 || F <= 0 || N <= 0 || ldx < F) {
        ign_set_error("ign_head_fwd: null pointer or bad dimension (B=%d F=%d N=%d ldx=%lld)", B, F, N, ldx);
        return IGN_E_ARG;
    }
    if (N > HEAD_NMAX) { ign_set_error("ign_head_fwd: N=%d classes > %d", N, HEAD_NMAX); return IGN_E_UNSUP; }
    if ((ldx & 3) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15) || (F & 3)) {
        ign_set_error("ign_head_fwd: X/W must be 16-byte aligned with F and ldx multiples of 4 (F=%d ldx=%lld)", F, ldx);
        return IGN_E_ARG;
    }
    IgnScopedTimer tm("head_fwd", (hipStream_t)stream);
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B), dim3(F >= 32768 ? 1024 : 256), 0, (hipStream_t)stream, X, W, bias, out, B, F, N, ldx);
    return ign_check_launch("head_fwd_kernel");
}

extern "C" int ign_head_bwd(const float* g, const float* X, const float* W, float* gX, float* gW, float* gbias, int B,
                            int F, int N, long long ldx, void* stream) {
    return ign_head_bwd_acc(g, X, W, gX, gW, gbias, nullptr, nullptr, B, F, N, ldx, stream);
}

extern "C" int ign_head_bwd_acc(const float* g, const float* X, const float* W, float* gX, float* gW, float* gbias,
                                const float* gW_add, const float* add_scale_dev, int B, int F, int N, long long ldx, void* stream) {
    if (!g || !X || !W || B <= 0 || F <= 0 || N <= 0 || ldx < F) {
        ign_set_error("ign_head_bwd: null pointer or bad dimension");
        return IGN_E_ARG;
    }
    if (N > HEAD_NMAX) { ign_set_error("ign_head_bwd: N=%d classes > %d", N, HEAD_NMAX); return IGN_E_UNSUP; }
    if ((ldx & 3) || (F & 3) || ((uintptr_t)W & 15) || (gX && ((uintptr_t)gX & 15))) {
        ign_set_error("ign_head_bwd: W/gX must be 16-byte aligned with F and ldx multiples of 4");
        return IGN_E_ARG;
    }
    if ((size_t)B * N * 4 > 40 * 1024) { ign_set_error("ign_head_bwd: B*N=%d too large for the LDS copy of g", B * N); return IGN_E_TOOBIG; }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (gX && gW) {
        const int nwb = (F + 31) / 32, nxb = (F / 4 + 255) / 256;
        IgnScopedTimer tm("head_bwd_xw", s);
        hipLaunchKernelGGL(head_bwd_xw_kernel, dim3((unsigned)(nwb + nxb * B)), dim3(256), ((size_t)B * N + 8 * HEAD_NMAX * 32) * 4, s, g, X,
                           W, gX, gW, gbias, B, F, N, ldx, gW_add, add_scale_dev, nwb, nxb);
        return ign_check_launch("head_bwd_xw_kernel");
    }
    if (gX) {
        IgnScopedTimer tm("head_bwd_x", s);
        hipLaunchKernelGGL(head_bwd_x_kernel, dim3((F / 4 + 255) / 256, B), dim3(256), 0, s, g, W, gX, B, F, N, ldx);
        if ((rc = ign_check_launch("head_bwd_x_kernel"))) return rc;
    }
    if (gW) {
        IgnScopedTimer tm("head_bwd_w", s);
        hipLaunchKernelGGL(head_bwd_w_kernel, dim3((F + 31) / 32), dim3(256), ((size_t)B * N + 8 * HEAD_NMAX * 32) * 4, s, g, X, gW, gbias, B, F, N, ldx,
                           gW_add, add_scale_dev);
        if ((rc = ign_check_launch("head_bwd_w_kernel"))) return rc;
    }
    return 0;
}

extern "C" int ign_gate_fwd(const float* sbm, const float* dnn, float* out, float* eta, int B, int N, float gating_value,
                            int use_gating_value, void* stream) {
    if (!sbm || !dnn || !out || !eta || B <= 0 || N < 2) {
        ign_set_error("ign_gate_fwd: null pointer or bad dimension (B=%d N=%d)", B, N);
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(gate_fwd_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, sbm, dnn, out, eta, B, N,
                       gating_value, use_gating_value);
    return ign_check_launch("gate_fwd_kernel");
}

extern "C" int ign_gate_bwd(const float* sbm, const float* dnn, const float* gout, const float* geta, float* gsbm,
                            float* gdnn, int B, int N, float gating_value, int use_gating_value, void* stream) {
    if (!sbm || !dnn || !gout || !gsbm || !gdnn || B <= 0 || N < 2) {
        ign_set_error("ign_gate_bwd: null pointer or bad dimension (B=%d N=%d)", B, N);
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(gate_bwd_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, sbm, dnn, gout, geta, gsbm,
                       gdnn, B, N, gating_value, use_gating_value);
    return ign_check_launch("gate_bwd_kernel");
}

extern "C" int ign_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                             float eps, int step, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step <= 0) {
        ign_set_error("ign_adam_step: null pointer, n <= 0 or step <= 0");
        return IGN_E_ARG;
    }
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) {
        ign_set_error("ign_adam_step: buffers must be 16-byte aligned");
        return IGN_E_ARG;
    }
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    IgnScopedTimer tm("adam", (hipStream_t)stream);
    const long long blocks = (n / 4 + 256) / 256;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                       eps, (float)bc1, (float)sqrt(bc2));
    return ign_check_launch("adam_kernel");
}

extern "C" int ign_gather_flat(const void* const* src, const long long* off, const long long* n, int count, float* flat,
                               void* stream) {
    if (!src || !off || !n || !flat || count < 0) {
        ign_set_error("ign_gather_flat: null pointer or negative count");
        return IGN_E_ARG;
    }
    for (int base = 0; base < count; base += GATHER_MAX) {
        GatherTable t;
        const int m = count - base < GATHER_MAX ? count - base : GATHER_MAX;
        long long big = 1;
        for (int i = 0; i < m; ++i) {
            t.src[i] = (const float*)src[base + i]; t.off[i] = off[base + i]; t.n[i] = n[base + i];
            if (!t.src[i] || t.n[i] < 0) { ign_set_error("ign_gather_flat: entry %d is null / negative", base + i); return IGN_E_ARG; }
            if (t.n[i] > big) big = t.n[i];
        }
        const long long bx = (big + 256 * 8 - 1) / (256 * 8);            // ~8 elements per thread for the largest entry
        hipLaunchKernelGGL(gather_flat_kernel, dim3((unsigned)(bx < 1 ? 1 : (bx > 1024 ? 1024 : bx)), (unsigned)m), dim3(256), 0,
                           (hipStream_t)stream, t, flat);
        int rc;
        if ((rc = ign_check_launch("gather_flat_kernel"))) return rc;
    }
    return 0;
}

extern "C" int ign_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                                 float eps, int* step_dev, float* bc_dev, void* stream) {
    if (!p || !g || !m || !v || !step_dev || !bc_dev || n <= 0) {
        ign_set_error("ign_adam_step_dev: null pointer or n <= 0");
        return IGN_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, s, step_dev, bc_dev, beta1, beta2);
    int rc;
    if ((rc = ign_check_launch("adam_tick_kernel"))) return rc;
    IgnScopedTimer tm("adam", s);
    const long long blocks = (n / 4 + 256) / 256;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, bc_dev);
    return ign_check_launch("adam_dev_kernel");
}

// ------------------------------------------------------------------------------------------------ shapelet diversity
// loss = mean_{c,i,j} exp(-|| w[i,c,:] - w[j,c,:] + 1e-6 ||_2) (1 - delta_ij)      IGN/model/Shapelet.py:223-230
// (nn.PairwiseDistance(p=2): the eps is added to the DIFFERENCE, so d_ij != d_ji in the last bits -- both are kept).
// One block per channel: pass 1 reduces the K*K squared distances into LDS, pass 2 forms the loss partial and the
// gradient of the (unit-weighted) loss w.r.t. w.  Replaces ~25 tiny elementwise / reduction launches per group.
constexpr int DIV_KMAX = 16;

// one block = one channel c of one group; thread 0 returns the loss partial (already times `scale`), gw receives scale * gradient
__device__ __forceinline__ float diversity_block(const float* __restrict__ w, float* __restrict__ gw, int K, int C, int L, int c,
                                                 float eps, float scale) {
    __shared__ float D2[DIV_KMAX][DIV_KMAX];        // D2[i][j] = sum_l (w_i - w_j + eps)^2
    __shared__ float Ew[DIV_KMAX][DIV_KMAX];        // exp(-D_ij) / D_ij   (0 on the diagonal)
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t cs = (size_t)C * L;                 // stride between shapelets
    const float* wc = w + (size_t)c * L;
    for (int i = 0; i < K; ++i)
        for (int j = i + 1; j < K; ++j) {
            float a = 0.f, b = 0.f;
            for (int l = tid; l < L; l += 256) {
                const float dl = wc[i * cs + l] - wc[j * cs + l];
                a = fmaf(dl + eps, dl + eps, a);
                b = fmaf(eps - dl, eps - dl, b);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            __syncthreads();
            if (lane == 0) red[wave] = a;
            __syncthreads();
            const float ta = red[0] + red[1] + red[2] + red[3];
            __syncthreads();
            if (lane == 0) red[wave] = b;
            __syncthreads();
            const float tb = red[0] + red[1] + red[2] + red[3];
            if (tid == 0) { D2[i][j] = ta; D2[j][i] = tb; }
        }
    __syncthreads();
    const float norm = 1.f / ((float)C * K * K);
    if (tid < K * K) {
        const int i = tid / K, j = tid % K;
        float e = 0.f, ew = 0.f;
        if (i != j) {
            const float d = sqrtf(D2[i][j]);
            e = expf(-d);
            ew = e / d;
        }
        Ew[i][j] = ew;
        D2[i][j] = e;                                // reuse as e_ij for the loss sum below
    }
    __syncthreads();
    float lossp = 0.f;
    if (tid == 0) {
        float s = 0.f;
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) s += D2[i][j];
        lossp = s * norm * scale;
    }
    // d e_ij / d w_i[l] = -e_ij (dl + eps) / D_ij ;  d e_ji / d w_i[l] = +e_ji (eps - dl) / D_ji   (dl = w_i[l] - w_j[l])
    for (int l = tid; l < L; l += 256)
        for (int i = 0; i < K; ++i) {
            const float wi = wc[i * cs + l];
            float g = 0.f;
            for (int j = 0; j < K; ++j)
                if (j != i) {
                    const float dl = wi - wc[j * cs + l];
                    g += -Ew[i][j] * (dl + eps) + Ew[j][i] * (eps - dl);
                }
            gw[i * cs + (size_t)c * L + l] = g * norm * scale;
        }
    return lossp;
}

__global__ void __launch_bounds__(256) diversity_kernel(const float* __restrict__ w, float* __restrict__ loss_part,
                                                        float* __restrict__ gw, int K, int C, int L, float eps) {
    const float lp = diversity_block(w, gw, K, C, L, blockIdx.x, eps, 1.f);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = lp;
}

// ------------------------------------------------------------------------------------------------ both SBM regularisers, one launch
// loss = lambda_reg * mean |W| + lambda_div * sum_g diversity_g   (ShapeBottleneckModel.loss, IGN/model/Shapelet.py:217-230) with
// the gradient of every term -- the value does not depend on the batch, so forward and backward are one launch; the consumers
// (head weight gradient, shapelet gradient reduction) add `upstream * gradient` in their own epilogues.
//   blocks [0, G*C): channel c of group g (diversity_block);  blocks [G*C, G*C + nwb): 1024 elements of W each.
// Every block leaves one partial; the block that draws the last ticket adds them up in a fixed order (thread-strided sums, then
// a fixed tree), so the result is bitwise reproducible although the arrival order is not.  Cross-block visibility follows the
// producer / consumer recipe of the MI355X guide (agent-scope release before the ticket, acquire after it, sc1 loads).
constexpr int REG_GMAX = 8;
struct RegTable {
    const float* w[REG_GMAX];
    float* gw[REG_GMAX];
    int K[REG_GMAX], L[REG_GMAX];
};
__global__ void __launch_bounds__(256) sbm_reg_kernel(const RegTable t, int G, int C, const float* __restrict__ W,
                                                      float* __restrict__ gWreg, long long nW, int nwb, float lam_reg, float lam_div,
                                                      float eps, float* __restrict__ parts, unsigned int* __restrict__ ticket,
                                                      float* __restrict__ loss_out) {
    __shared__ float red[256];
    __shared__ int is_last;
    const int blk = blockIdx.x, tid = threadIdx.x;
    const int nblk = G * C + nwb;
    float lp = 0.f;
    if (blk < G * C) {
        const int g = blk / C, c = blk - g * C;
        lp = diversity_block(t.w[g], t.gw[g], t.K[g], C, t.L[g], c, eps, lam_div);
    } else {
        // lambda_reg * mean |W|: gradient lambda_reg * sign(W) / numel (sign(0) = 0, as aten::sgn)
        const long long i0 = (long long)(blk - G * C) * 1024 + tid * 4;
        const float sc = lam_reg / (float)nW;
        float a = 0.f;
        for (int u = 0; u < 4; ++u) {
            const long long i = i0 + u;
            if (i < nW) {
                const float v = W[i];
                a += fabsf(v);
                gWreg[i] = v > 0.f ? sc : (v < 0.f ? -sc : 0.f);
            }
        }
        red[tid] = a;
        __syncthreads();
        if (tid == 0) {
            float sacc = 0.f;
            for (int i = 0; i < 256; ++i) sacc += red[i];
            lp = sacc * sc;
        }
    }
    if (tid == 0) {
        __hip_atomic_store(parts + blk, lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (old == (unsigned int)(nblk - 1));
        if (is_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!is_last) return;
    float a = 0.f;
    for (int i = tid; i < nblk; i += 256) a += __hip_atomic_load(parts + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    red[tid] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        loss_out[0] = red[0];
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch on this stream
    }
}

extern "C" size_t ign_sbm_reg_workspace_bytes(int G, int C, long long nW) {
    if (G < 0 || G > REG_GMAX || C <= 0 || nW < 0) return 0;
    return (size_t)(G * C + (nW + 1023) / 1024 + 4) * sizeof(float);
}

extern "C" int ign_sbm_reg_fwd_bwd(const float* W, float* gW_reg, long long nW, float lambda_reg, int G, const float* const* w_kcl,
                                   float* const* gw_kcl, const int* K, const int* L, int C, float lambda_div, float eps,
                                   float* loss_out, void* workspace, void* stream) {
    static const char* who = "ign_sbm_reg_fwd_bwd";
    if (!loss_out || !workspace || G < 0 || G > REG_GMAX || C <= 0 || nW < 0 || (nW > 0 && (!W || !gW_reg)) ||
        (G > 0 && (!w_kcl || !gw_kcl || !K || !L))) {
        ign_set_error("%s: null pointer or bad dimension (G=%d C=%d nW=%lld)", who, G, C, nW);
        return IGN_E_ARG;
    }
    RegTable t;
    for (int g = 0; g < G; ++g) {
        if (!w_kcl[g] || !gw_kcl[g] || K[g] <= 0 || L[g] <= 0) { ign_set_error("%s: group %d: null pointer or bad K / L", who, g); return IGN_E_ARG; }
        if (K[g] > DIV_KMAX) { ign_set_error("%s: K=%d > %d shapelets per group", who, K[g], DIV_KMAX); return IGN_E_UNSUP; }
        t.w[g] = w_kcl[g]; t.gw[g] = gw_kcl[g]; t.K[g] = K[g]; t.L[g] = L[g];
    }
    const int nwb = (int)((nW + 1023) / 1024);
    const int nblk = G * C + nwb;
    if (nblk <= 0) { ign_set_error("%s: nothing to do", who); return IGN_E_ARG; }
    float* parts = (float*)workspace;
    unsigned int* ticket = (unsigned int*)(parts + nblk);      // the caller zero-fills the workspace ONCE; the kernel re-arms it
    IgnScopedTimer tm("sbm_reg", (hipStream_t)stream);
    hipLaunchKernelGGL(sbm_reg_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, t, G, C, W, gW_reg, nW, nwb, lambda_reg,
                       lambda_div, eps, parts, ticket, loss_out);
    return ign_check_launch("sbm_reg_kernel");
}

extern "C" int ign_diversity_fwd_bwd(const float* w_kcl, float* loss_part_c, float* gw_kcl, int K, int C, int L, float eps,
                                     void* stream) {
    if (!w_kcl || !loss_part_c || !gw_kcl || K <= 0 || C <= 0 || L <= 0) {
        ign_set_error("ign_diversity_fwd_bwd: null pointer or bad dimension (K=%d C=%d L=%d)", K, C, L);
        return IGN_E_ARG;
    }
    if (K > DIV_KMAX) { ign_set_error("ign_diversity_fwd_bwd: K=%d > %d shapelets per group", K, DIV_KMAX); return IGN_E_UNSUP; }
    IgnScopedTimer tm("diversity", (hipStream_t)stream);
    hipLaunchKernelGGL(diversity_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, w_kcl, loss_part_c, gw_kcl, K, C, L, eps);
    return ign_check_launch("diversity_kernel");
}

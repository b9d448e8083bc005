// BatchNorm-1 of the EEG-CNN block from data statistics (IGN/model/eegcnn.py:90-91; models/eegcnn.py::_forward_hip).
// The batch mean and variance of y[f] = w1[f] (*) x over (batch, electrode, time) are a linear and a quadratic form in the filter:
//     mean_f = w_f . S / n,      E[y_f^2] = w_f^T G w_f / n,
// S[j] = sum of the samples tap j sees, G[j][j'] = sum_{rows,t} xp[t+j] xp[t+j'] (xp = the zero-padded row) -- both from one pass
// over the data (ign_autocorr_fwd, ign_edge_lagprod_fwd).  Rounds 1-3 assembled G and the forms with float64 torch ops (cumsum,
// flip, cat, gather, einsum + autograd: ~45 launches of a few microseconds each per step); these three kernels do the same
// arithmetic in float64 in three launches.  Tiny problems (k <= 125 taps, F1 = 8 filters): latency, not throughput.
#include "ign_common.h"

constexpr int B1_MAXM = 124, B1_LD = 128;           // layout of the summed edge partials: (2, 124, 128), column 127 = column sums

// Float64 sums of the fp32 per-block partials of the two data passes, one launch:
//   blocks [0, nbD):        D[col] = sum_b pe[b][col], col < 2*124*128 (thread <-> column: coalesced; 8 loads in flight)
//   blocks [nbD, nbD+nsC):  slice s of the autocorrelation partials: Cs[s][d] = sum over its rows of pc[.][d], ts[s] = sum of prs[.]
// (the slices are added by bn1_gram_kernel in ascending order: fixed order, reproducible).
constexpr int B1_CSLICE = 64;                       // autocorrelation partial rows per slice
__global__ void __launch_bounds__(256) bn1_partial_sums_kernel(const float* __restrict__ pe, int npe, const float* __restrict__ pc,
                                                               const float* __restrict__ prs, int npc, int k, int nbD,
                                                               double* __restrict__ D, double* __restrict__ Cs,
                                                               double* __restrict__ ts) {
    constexpr int NCOL = 2 * B1_MAXM * B1_LD;
    if ((int)blockIdx.x < nbD) {
        const int col = blockIdx.x * 256 + threadIdx.x;
        if (col >= NCOL) return;
        double s = 0.0;
        int b = 0;
        for (; b + 8 <= npe; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pe[(size_t)(b + u) * NCOL + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; b < npe; ++b) s += (double)pe[(size_t)b * NCOL + col];
        D[col] = s;
        return;
    }
    const int sl = blockIdx.x - nbD, d = threadIdx.x;
    const int r0 = sl * B1_CSLICE, r1 = min(npc, r0 + B1_CSLICE);
    if (d < k) {
        double s = 0.0;
        for (int r = r0; r < r1; ++r) s += (double)pc[(size_t)r * k + d];
        Cs[(size_t)sl * B1_LD + d] = s;
    } else if (d == 128) {
        double s = 0.0;
        for (int r = r0; r < r1; ++r) s += (double)prs[r];
        ts[sl] = s;
    }
}

// One block, thread d < k walks the d-th diagonal of G with running prefix / suffix sums of the edge terms.
__global__ void __launch_bounds__(128) bn1_gram_kernel(const double* __restrict__ Cs, const double* __restrict__ ts, int nsC,
                                                       const double* __restrict__ D, double* __restrict__ G,
                                                       double* __restrict__ S, int k, int pl) {
    const int d = threadIdx.x, m = k - 1;
    const double* Dh = D;
    const double* Dt = D + (size_t)B1_MAXM * B1_LD;
    if (d < k) {
        double c = 0.0, total = 0.0;
        for (int sl = 0; sl < nsC; ++sl) { c += Cs[(size_t)sl * B1_LD + d]; total += ts[sl]; }
        // G[j][j+d] = C[d] - sum_{s<j} Dh[s][d] - sum_{s>=j} Dt[s][d],  j = 0 .. k-1-d   (entries with s + d >= m are zero by construction)
        // (loads in batches of 8 before they are used: a single block walking 125 dependent steps is pure latency otherwise)
        double pt = 0.0;
        for (int s0 = 0; s0 < m; s0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (s0 + u < m) ? Dt[(size_t)(s0 + u) * B1_LD + d] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) pt += v[u];
        }
        double ph = 0.0;
        for (int j0 = 0; j0 + d < k; j0 += 8) {
            double vh[8], vt[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                vh[u] = (j < m) ? Dh[(size_t)j * B1_LD + d] : 0.0;
                vt[u] = (j < m) ? Dt[(size_t)j * B1_LD + d] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j + d < k) {
                    const double g = c - ph - pt;
                    G[(size_t)j * k + j + d] = g;
                    G[(size_t)(j + d) * k + j] = g;
                    ph += vh[u]; pt -= vt[u];
                }
            }
        }
        // S[j]: all samples minus those tap j never sees.  Column 127 holds ch[i] = sum_rows xp[i] and ct[i] = sum_rows xp[T+i];
        // x[t] = xp[pl + t].  Tap j < pl misses the LAST pl - j samples, tap j > pl the FIRST j - pl.
        const int j = d;
        double miss = 0.0;
        if (j < pl)      for (int i = j; i < pl; ++i) miss += Dt[(size_t)i * B1_LD + 127];
        else if (j > pl) for (int t = 0; t < j - pl; ++t) miss += Dh[(size_t)(pl + t) * B1_LD + 127];
        S[j] = total - miss;
    }
}

__device__ __forceinline__ double block_sum_128(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double t = sh[0] + sh[1];
    __syncthreads();
    return t;
}

// saved[f][0..k) = u = G w_f;  saved[f][k..k+4) = mu, var, r = 1/sqrt(var + eps) (as the float the forward used), b1
__global__ void __launch_bounds__(128) bn1_fold_fwd_kernel(const float* __restrict__ w1, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ rs,
                                                           const double* __restrict__ G, const double* __restrict__ S, double n,
                                                           float eps, float momentum, float* __restrict__ run_mean,
                                                           float* __restrict__ run_var, float* __restrict__ alpha,
                                                           float* __restrict__ cshift, double* __restrict__ saved, int k, int Dm) {
    __shared__ double sh[2];
    __shared__ float wf[128];
    const int f = blockIdx.x, j = threadIdx.x;
    wf[j] = j < k ? w1[(size_t)f * k + j] : 0.f;
    __syncthreads();
    double u = 0.0;
    if (j < k)
        for (int jp = 0; jp < k; ++jp) u = fma(G[(size_t)jp * k + j], (double)wf[jp], u);      // G symmetric: column j, coalesced
    const double q = block_sum_128(j < k ? (double)wf[j] * u : 0.0, sh);
    const double ms = block_sum_128(j < k ? (double)wf[j] * S[j] : 0.0, sh);
    const float mu1 = (float)(ms / n);                                 // the float the rest of the network sees
    const float var1 = (float)(q / n - (double)mu1 * (double)mu1);
    const float r = 1.f / sqrtf(var1 + eps);
    const float a1 = gamma[f] * r;
    const float b1 = beta[f] - a1 * mu1;
    double* sv = saved + (size_t)f * (k + 4);
    if (j < k) sv[j] = u;
    if (j == 0) {
        sv[k] = (double)mu1; sv[k + 1] = (double)var1; sv[k + 2] = (double)r; sv[k + 3] = (double)b1;
        if (run_mean) {
            run_mean[f] = (1.f - momentum) * run_mean[f] + momentum * mu1;
            run_var[f] = (1.f - momentum) * run_var[f] + momentum * (float)((double)var1 * (n / (n - 1.0)));
        }
    }
    for (int i = j; i < Dm; i += 128) {
        alpha[(size_t)f * Dm + i] = a1;
        cshift[(size_t)f * Dm + i] = b1 * rs[(size_t)f * Dm + i];
    }
}

__global__ void __launch_bounds__(128) bn1_fold_bwd_kernel(const float* __restrict__ g_alpha, const float* __restrict__ g_cshift,
                                                           const float* __restrict__ gamma, const float* __restrict__ rs,
                                                           const double* __restrict__ S, const double* __restrict__ saved,
                                                           double n, float* __restrict__ g_w1, float* __restrict__ g_gamma,
                                                           float* __restrict__ g_beta, float* __restrict__ g_rs, int k, int Dm) {
    __shared__ double sh[2];
    const int f = blockIdx.x, j = threadIdx.x;
    const double* sv = saved + (size_t)f * (k + 4);
    const double mu = sv[k], r = sv[k + 2], b1 = sv[k + 3];
    double ga = 0.0, gb = 0.0;
    for (int i = j; i < Dm; i += 128) {
        const double gc = (double)g_cshift[(size_t)f * Dm + i];
        ga += (double)g_alpha[(size_t)f * Dm + i];
        gb += gc * (double)rs[(size_t)f * Dm + i];
        g_rs[(size_t)f * Dm + i] = (float)(gc * b1);
    }
    ga = block_sum_128(ga, sh);
    gb = block_sum_128(gb, sh);
    // a = gamma r, b = beta - a mu, r = (var + eps)^(-1/2)
    const double gam = (double)gamma[f];
    const double g_a = ga - gb * mu;                   // dL/da including the path through b
    const double g_mu = -gb * gam * r;
    const double g_var = g_a * gam * (-0.5 * r * r * r);
    if (j == 0) { g_gamma[f] = (float)(g_a * r); g_beta[f] = (float)gb; }
    if (j < k) {
        const double Sj = S[j];
        g_w1[(size_t)f * k + j] = (float)(g_mu * Sj / n + g_var * (2.0 * sv[j] / n - 2.0 * mu * Sj / n));
    }
}

// workspace: autocorrelation partials + row sums, edge partials (fp32), then the float64 sums
struct Bn1Ws { size_t pc, prs, pe, D, Cs, ts, total; int npc_max, npe, nsC_max; };
static Bn1Ws bn1_ws_layout(int rows, int k) {
    Bn1Ws w;
    w.npc_max = (int)ign_autocorr_parts(rows);
    w.npe = (int)ign_edge_lagprod_parts(rows);
    w.nsC_max = (w.npc_max + B1_CSLICE - 1) / B1_CSLICE;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    w.pc = take((size_t)w.npc_max * k * sizeof(float));
    w.prs = take((size_t)w.npc_max * sizeof(float));
    w.pe = take((size_t)w.npe * 2 * B1_MAXM * B1_LD * sizeof(float));
    w.D = take((size_t)2 * B1_MAXM * B1_LD * sizeof(double));
    w.Cs = take((size_t)w.nsC_max * B1_LD * sizeof(double));
    w.ts = take((size_t)w.nsC_max * sizeof(double));
    w.total = o;
    return w;
}

extern "C" size_t ign_bn1_data_stats_workspace_bytes(int rows, int T, int k) {
    if (rows <= 0 || T <= 0 || k < 2 || k - 1 > B1_MAXM) return 0;
    return bn1_ws_layout(rows, k).total;
}

extern "C" int ign_bn1_data_stats(const float* x_rows, int rows, int T, int k, int pad_left, void* workspace, double* G, double* S,
                                  void* stream) {
    if (!x_rows || !workspace || !G || !S || rows <= 0 || T < k || k < 2 || k - 1 > B1_MAXM || pad_left < 0 || pad_left >= k
        || ((uintptr_t)workspace & 15)) {
        ign_set_error("ign_bn1_data_stats: null / unaligned pointer or bad shape (rows=%d T=%d k=%d pad_left=%d; needs 2 <= k <= %d, T >= k)",
                      rows, T, k, pad_left, B1_MAXM + 1);
        return IGN_E_ARG;
    }
    const Bn1Ws w = bn1_ws_layout(rows, k);
    char* base = (char*)workspace;
    float *pc = (float*)(base + w.pc), *prs = (float*)(base + w.prs), *pe = (float*)(base + w.pe);
    double *D = (double*)(base + w.D), *Cs = (double*)(base + w.Cs), *ts = (double*)(base + w.ts);
    int rc, npc = 0;
    if ((rc = ign_autocorr_sum_fwd(x_rows, pc, prs, rows, T, k, &npc, stream))) return rc;        // IGN_E_UNSUP beyond T = 1024
    if ((rc = ign_edge_lagprod_fwd(x_rows, pe, rows, T, k, pad_left, stream))) return rc;
    const int nsC = (npc + B1_CSLICE - 1) / B1_CSLICE;
    const int nbD = (2 * B1_MAXM * B1_LD + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn1_partial_sums_kernel, dim3(nbD + nsC), dim3(256), 0, s, pe, w.npe, pc, prs, npc, k, nbD, D, Cs, ts);
    if ((rc = ign_check_launch("bn1_partial_sums_kernel"))) return rc;
    hipLaunchKernelGGL(bn1_gram_kernel, dim3(1), dim3(128), 0, s, Cs, ts, nsC, D, G, S, k, pad_left);
    return ign_check_launch("bn1_gram_kernel");
}

extern "C" int ign_bn1_fold_fwd(const float* w1, const float* gamma, const float* beta, const float* rs, const double* G,
                                const double* S, double n, float eps, float momentum, float* running_mean, float* running_var,
                                float* alpha, float* cshift, double* saved, int F1, int k, int Dm, void* stream) {
    if (!w1 || !gamma || !beta || !rs || !G || !S || !alpha || !cshift || !saved || F1 <= 0 || k < 2 || k > 128 || Dm <= 0
        || (long long)F1 * Dm > 4096 || n <= 1.0 || (running_mean != nullptr) != (running_var != nullptr)) {
        ign_set_error("ign_bn1_fold_fwd: null pointer or bad shape (F1=%d k=%d D=%d n=%g)", F1, k, Dm, n);
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(bn1_fold_fwd_kernel, dim3(F1), dim3(128), 0, (hipStream_t)stream, w1, gamma, beta, rs, G, S, n, eps, momentum,
                       running_mean, running_var, alpha, cshift, saved, k, Dm);
    return ign_check_launch("bn1_fold_fwd_kernel");
}

extern "C" int ign_bn1_fold_bwd(const float* g_alpha, const float* g_cshift, const float* w1, const float* gamma, const float* rs,
                                const double* S, const double* saved, double n, float* g_w1, float* g_gamma, float* g_beta,
                                float* g_rs, int F1, int k, int Dm, void* stream) {
    (void)w1;
    if (!g_alpha || !g_cshift || !gamma || !rs || !S || !saved || !g_w1 || !g_gamma || !g_beta || !g_rs || F1 <= 0 || k < 2 || k > 128
        || Dm <= 0 || (long long)F1 * Dm > 4096 || n <= 1.0) {
        ign_set_error("ign_bn1_fold_bwd: null pointer or bad shape (F1=%d k=%d D=%d n=%g)", F1, k, Dm, n);
        return IGN_E_ARG;
    }
    hipLaunchKernelGGL(bn1_fold_bwd_kernel, dim3(F1), dim3(128), 0, (hipStream_t)stream, g_alpha, g_cshift, gamma, rs, S, saved, n, g_w1,
                       g_gamma, g_beta, g_rs, k, Dm);
    return ign_check_launch("bn1_fold_bwd_kernel");
}

// Channels-last 1-D convolution as an implicit GEMM -- the fp32-MFMA kernels (v_mfma_f32_32x32x2_f32) and the C entry points of the forward / data-gradient / weight-gradient GEMMs (the split-bf16 kernels they can dispatch to live in ign_clconv_x6.hip), with the
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
#include "ign_clconv.h"

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

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nsplit,
                                                           int Co, int Ci, int k) {
    const long long n = (long long)Co * Ci * k;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // index in (Co, k, Ci) order: coalesced reads
    if (i >= n) return;
    float s = 0.f;
    int p = 0;
    for (; p + 8 <= nsplit; p += 8) {            // 8 loads in flight, summed in ascending order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long long)(p + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < nsplit; ++p) s += part[(long long)p * n + i];
    const int ci = (int)(i % Ci);
    const long long t = i / Ci;
    const int j = (int)(t % k);
    const long long co = t / k;
    dw[(co * Ci + ci) * k + j] = s;
}

// the reductions of several layers in one launch (blockIdx.y = layer); same arithmetic as wgrad_reduce_kernel
constexpr int WRED_LMAX = 8;
struct WgradReduceTable {
    const float* part[WRED_LMAX];
    float* dw[WRED_LMAX];
    int nsplit[WRED_LMAX], Co[WRED_LMAX], Ci[WRED_LMAX], k[WRED_LMAX];
};
__global__ void __launch_bounds__(256) wgrad_reduce_multi_kernel(const WgradReduceTable t) {
    const int l = blockIdx.y;
    const int Co = t.Co[l], Ci = t.Ci[l], k = t.k[l], nsplit = t.nsplit[l];
    const float* __restrict__ part = t.part[l];
    const long long n = (long long)Co * Ci * k;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    int p = 0;
    for (; p + 8 <= nsplit; p += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(long long)(p + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < nsplit; ++p) s += part[(long long)p * n + i];
    const int ci = (int)(i % Ci);
    const long long tt = i / Ci;
    const int j = (int)(tt % k);
    const long long co = tt / k;
    t.dw[l][(co * Ci + ci) * k + j] = s;
}

extern "C" int ign_clconv_wgrad_reduce_multi(int n, const void* const* part, float* const* dw_oik, const int* nsplit, const int* Co,
                                             const int* Ci, const int* k, void* stream) {
    if (n <= 0 || n > WRED_LMAX || !part || !dw_oik || !nsplit || !Co || !Ci || !k) {
        ign_set_error("ign_clconv_wgrad_reduce_multi: n=%d outside 1..%d or null table", n, WRED_LMAX);
        return IGN_E_ARG;
    }
    WgradReduceTable t;
    long long nmax = 0;
    for (int l = 0; l < n; ++l) {
        if (!part[l] || !dw_oik[l] || nsplit[l] <= 0 || Co[l] <= 0 || Ci[l] <= 0 || k[l] <= 0) {
            ign_set_error("ign_clconv_wgrad_reduce_multi: layer %d: bad argument", l);
            return IGN_E_ARG;
        }
        t.part[l] = (const float*)part[l]; t.dw[l] = dw_oik[l]; t.nsplit[l] = nsplit[l]; t.Co[l] = Co[l]; t.Ci[l] = Ci[l]; t.k[l] = k[l];
        const long long e = (long long)Co[l] * Ci[l] * k[l];
        nmax = e > nmax ? e : nmax;
    }
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    return ign_check_launch("wgrad_reduce_multi_kernel");
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

// ------------------------------------------------------------------------------------------------ C ABI
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

// x6: 0 = fp32 MFMA, 6 = split bf16 (six products), 1 = operands rounded to bf16 (one product)
static int clconv_fwd_impl(const char* who, int x6, const float* x, const void* wt, const float* bias, const float* pro_a,
                           const float* pro_b, float* y, float* stat_part, int B, int Tin, int Ci, int Co, int k, void* stream,
                           const float* bound_a = nullptr, const float* bound_w = nullptr, float* amax_out = nullptr,
                           int epi = EPI_BIAS_STATS, const float* ey = nullptr) {
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
    a.B3 = x6 ? (const unsigned short*)wt : nullptr; a.Kp = 0;
    a.pro_a = pro_a; a.pro_b = pro_b; a.pro_c = Ci; a.part = stat_part; a.amax_out = amax_out; a.ey = ey;
    a.mtiles = (int)((M + TM - 1) / TM); a.ntiles = (Co + TN - 1) / TN;
    IgnScopedTimer tm("clconv_fwd", (hipStream_t)stream);
    if (x6) {
        if (k > 16) { ign_set_error("%s: k=%d > 16 taps", who, k); return IGN_E_UNSUP; }
        ConvX6Args c{};
        c.g = a;
        c.cin = Ci; c.cp = (Ci + 15) / 16 * 16; c.k = k; c.g.Kp = k * c.cp;
        c.sample_pitch = (long long)Tin * Ci; c.rows_in = Tin; c.trows = Tout; c.tps = (Tout + TM - 1) / TM;
        c.g.mtiles = B * c.tps; c.nprod = x6; c.bound_a = bound_a; c.bound_b = bound_w;
        return ign_clconv_launch_x6t(c, epi, ign_vec_width(Ci), pro_a != nullptr, (hipStream_t)stream);
    }
    if (epi != EPI_BIAS_STATS) { ign_set_error("%s: epilogue %d needs the split kernels", who, epi); return IGN_E_UNSUP; }
    return launch_nt<EPI_BIAS_STATS>(a, ign_vec_width(Ci), pro_a != nullptr, (hipStream_t)stream);
}

extern "C" int ign_clconv_fwd(const float* x, const float* wt, const float* bias, const float* pro_a, const float* pro_b,
                              float* y, float* stat_part, int B, int Tin, int Ci, int Co, int k, void* stream) {
    return clconv_fwd_impl("ign_clconv_fwd", 0, x, wt, bias, pro_a, pro_b, y, stat_part, B, Tin, Ci, Co, k, stream);
}

extern "C" int ign_clconv_fwd_x6(const float* x, const void* wt3, const float* bias, const float* pro_a, const float* pro_b,
                                 float* y, float* stat_part, int B, int Tin, int Ci, int Co, int k, void* stream) {
    return clconv_fwd_impl("ign_clconv_fwd_x6", 6, x, wt3, bias, pro_a, pro_b, y, stat_part, B, Tin, Ci, Co, k, stream);
}

extern "C" int ign_clconv_fwd_bf16(const float* x, const void* wt3, const float* bias, const float* pro_a, const float* pro_b,
                                   float* y, float* stat_part, int B, int Tin, int Ci, int Co, int k, void* stream) {
    return clconv_fwd_impl("ign_clconv_fwd_bf16", 1, x, wt3, bias, pro_a, pro_b, y, stat_part, B, Tin, Ci, Co, k, stream);
}

extern "C" int ign_clconv_fwd_h3(const float* x, const void* wt_h2, const float* bias, const float* pro_a, const float* pro_b,
                                 float* y, float* stat_part, const float* bound_in, const float* bound_w, int B, int Tin, int Ci,
                                 int Co, int k, void* stream) {
    return ign_clconv_fwd_h3_amax(x, wt_h2, bias, pro_a, pro_b, y, stat_part, bound_in, bound_w, nullptr, B, Tin, Ci, Co, k, stream);
}

extern "C" int ign_clconv_fwd_h3_amax(const float* x, const void* wt_h2, const float* bias, const float* pro_a, const float* pro_b,
                                      float* y, float* stat_part, const float* bound_in, const float* bound_w, float* amax_out, int B,
                                      int Tin, int Ci, int Co, int k, void* stream) {
    if (!bound_in || !bound_w) { ign_set_error("ign_clconv_fwd_h3: null operand bound"); return IGN_E_ARG; }
    return clconv_fwd_impl("ign_clconv_fwd_h3", 3, x, wt_h2, bias, pro_a, pro_b, y, stat_part, B, Tin, Ci, Co, k, stream, bound_in,
                           bound_w, amax_out);
}

// du = (g W) * gelu'(u): the input gradient of a dense layer z = gelu(u) W^T + b, through the activation, in one GEMM
// (g (M, Co), wd_h2 = the transposed packed weight as for ign_clconv_fwd_h3's input gradient, u / du (M, Ci), Ci % 256 == 0).
extern "C" int ign_linear_dgrad_gelu_h3(const float* g, const void* wd_h2, const float* u, float* du, const float* bound_g,
                                        const float* bound_w, float* amax_out, long long M, int Co, int Ci, void* stream) {
    if (!bound_g || !bound_w || !u) { ign_set_error("ign_linear_dgrad_gelu_h3: null bound / pre-activation"); return IGN_E_ARG; }
    if (M <= 0 || M > 0x3fffffffLL || Ci % 256 || Co % 4) {
        ign_set_error("ign_linear_dgrad_gelu_h3: needs 0 < M < 2^30, Ci %% 256 == 0, Co %% 4 == 0 (M=%lld Co=%d Ci=%d)", M, Co, Ci);
        return IGN_E_UNSUP;
    }
    return clconv_fwd_impl("ign_linear_dgrad_gelu_h3", 3, g, wd_h2, nullptr, nullptr, nullptr, du, nullptr, 1, (int)M, Co, Ci, 1, stream,
                           bound_g, bound_w, amax_out, EPI_GELU_BWD, u);
}

static int clconv_dgrad_impl(const char* who, int x6, const float* dyp, const void* wt_dgrad, const float* y_in, const float* a_in,
                             const float* b_in, const float* mean_in, const float* invstd_in, float* g_in, float* stat_part, int B,
                             int Tin, int Ci, int Co, int k, void* stream, const float* bound_a = nullptr,
                             const float* bound_w = nullptr) {
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
    a.B3 = x6 ? (const unsigned short*)wt_dgrad : nullptr; a.Kp = 0;
    a.part = stat_part; a.ey = y_in; a.ea = a_in; a.eb = b_in; a.emean = mean_in; a.einv = invstd_in;
    a.mtiles = (int)((M + TM - 1) / TM); a.ntiles = (Ci + TN - 1) / TN;
    IgnScopedTimer tm("clconv_dgrad", (hipStream_t)stream);
    if (x6) {
        if (k > 16) { ign_set_error("%s: k=%d > 16 taps", who, k); return IGN_E_UNSUP; }
        ConvX6Args c{};
        c.g = a;
        c.cin = Co; c.cp = (Co + 15) / 16 * 16; c.k = k; c.g.Kp = k * c.cp;
        c.sample_pitch = (long long)(Tout + 2 * (k - 1)) * Co; c.rows_in = Tout + 2 * (k - 1); c.trows = Tin; c.tps = (Tin + TM - 1) / TM;
        c.g.mtiles = B * c.tps; c.nprod = x6; c.bound_a = bound_a; c.bound_b = bound_w;
        return ign_clconv_launch_x6t(c, EPI_MASK_STATS, ign_vec_width(Co), false, (hipStream_t)stream);
    }
    return launch_nt<EPI_MASK_STATS>(a, ign_vec_width(Co), false, (hipStream_t)stream);
}

extern "C" int ign_clconv_dgrad(const float* dyp, const float* wt_dgrad, const float* y_in, const float* a_in, const float* b_in,
                                const float* mean_in, const float* invstd_in, float* g_in, float* stat_part, int B, int Tin,
                                int Ci, int Co, int k, void* stream) {
    return clconv_dgrad_impl("ign_clconv_dgrad", 0, dyp, wt_dgrad, y_in, a_in, b_in, mean_in, invstd_in, g_in, stat_part, B, Tin,
                             Ci, Co, k, stream);
}

extern "C" int ign_clconv_dgrad_x6(const float* dyp, const void* wt3_dgrad, const float* y_in, const float* a_in, const float* b_in,
                                   const float* mean_in, const float* invstd_in, float* g_in, float* stat_part, int B, int Tin,
                                   int Ci, int Co, int k, void* stream) {
    return clconv_dgrad_impl("ign_clconv_dgrad_x6", 6, dyp, wt3_dgrad, y_in, a_in, b_in, mean_in, invstd_in, g_in, stat_part, B,
                             Tin, Ci, Co, k, stream);
}

extern "C" int ign_clconv_dgrad_bf16(const float* dyp, const void* wt3_dgrad, const float* y_in, const float* a_in, const float* b_in,
                                     const float* mean_in, const float* invstd_in, float* g_in, float* stat_part, int B, int Tin,
                                     int Ci, int Co, int k, void* stream) {
    return clconv_dgrad_impl("ign_clconv_dgrad_bf16", 1, dyp, wt3_dgrad, y_in, a_in, b_in, mean_in, invstd_in, g_in, stat_part, B,
                             Tin, Ci, Co, k, stream);
}

extern "C" int ign_clconv_dgrad_h3(const float* dyp, const void* wt_h2_dgrad, const float* y_in, const float* a_in, const float* b_in,
                                   const float* mean_in, const float* invstd_in, float* g_in, float* stat_part, const float* bound_dy,
                                   const float* bound_w, int B, int Tin, int Ci, int Co, int k, void* stream) {
    if (!bound_dy || !bound_w) { ign_set_error("ign_clconv_dgrad_h3: null operand bound"); return IGN_E_ARG; }
    return clconv_dgrad_impl("ign_clconv_dgrad_h3", 3, dyp, wt_h2_dgrad, y_in, a_in, b_in, mean_in, invstd_in, g_in, stat_part, B,
                             Tin, Ci, Co, k, stream, bound_dy, bound_w);
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
    const int V = ign_vec_width(Ci);
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
    return ign_clconv_launch_wgrad_reduce((const float*)workspace, dw_oik, a.nsplit, Co, Ci, k, s);
}

int ign_clconv_launch_wgrad_reduce(const float* part, float* dw_oik, int nsplit, int Co, int Ci, int k, hipStream_t s) {
    const long long n = (long long)Co * Ci * k;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, dw_oik, nsplit, Co, Ci, k);
    return ign_check_launch("wgrad_reduce_kernel");
}


// LayerNorm over the last dimension of an (R, D) matrix, forward and backward (IGN/layers/Transformer_EncDec.py:36-37,48,76-77;
// nn.TransformerEncoderLayer.norm1/2 of IGN/model/eegcnn.py:219-228; IGN/model/TimesNet.py:197): HBM-bound, 8 / 12 bytes per
// element.  A row is held by G = min(64, pow2ceil(D/4)) lanes as float4 pieces (so 64/G rows per wave: sixteen 16-float rows,
// four 64-float rows, one row of >= 256 floats), statistics are two passes over registers (mean, then centred second moment: no
// E[x^2]-E[x]^2 cancellation), reductions are xor-shuffles inside the G lanes.  The backward keeps per-lane partial sums of
// d(gamma), d(beta) over all rows a lane sees (its columns never change), folds them through LDS per block and a fixed-order
// second pass -- bitwise reproducible, no atomics.  torch's kernels for D = 64 run one row per block-sized unit: 3.9 / 6.3 ms
// per call at 3.9 M rows (PatchTST), where these take the HBM time.
#include <algorithm>
#include "ign_common.h"

constexpr int LN_MAXV = 16;                   // float4 pieces per lane: D <= 64 * 4 * 16 = 4096
constexpr int LN_ITERS_MAX = 32;              // row groups per wave (fewer when that would leave the chip under-filled)

struct LnArgs {
    const float *x, *gy, *gamma, *beta, *mean_in, *rstd_in;
    const float* res;                         // forward, nullable: the row normalised is x + res (residual connection) ...
    float* sum;                               // ... and x + res is written here (the backward's `x`)
    float *y, *gx, *mean, *rstd, *part;       // part: (nblocks, 2, D)
    float* amax;                              // backward, nullable: max |gx| as an atomic maximum (fp16 GEMM operand bound)
    long long R;
    int D, G, nv;                             // lanes per row, float4 pieces per lane
    int iters;                                // row groups per wave
    float eps;
};

__device__ __forceinline__ float group_sum(float v, int G) {
    for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int NV>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const LnArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = a.G, rpw = 64 / G, sub = lane / G, li = lane - sub * G;
    const int D4 = a.D >> 2;
    const float invD = 1.f / (float)a.D;
    float4 gm[NV], bt[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = li + v * G;
        gm[v] = c < D4 ? reinterpret_cast<const float4*>(a.gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        bt[v] = (c < D4 && a.beta) ? reinterpret_cast<const float4*>(a.beta)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * a.iters * rpw;
    for (int it = 0; it < a.iters; ++it) {
        const long long r = row0 + (long long)it * rpw + sub;
        if (row0 + (long long)it * rpw >= a.R) break;               // wave-uniform
        const bool ok = r < a.R;
        const long long rr = ok ? r : a.R - 1;
        const float4* xr = reinterpret_cast<const float4*>(a.x + rr * a.D);
        const float4* er = a.res ? reinterpret_cast<const float4*>(a.res + rr * a.D) : nullptr;
        float4 xv[NV];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = li + v * G;
            xv[v] = c < D4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (er && c < D4) {
                const float4 e = er[c];
                xv[v] = make_float4(xv[v].x + e.x, xv[v].y + e.y, xv[v].z + e.z, xv[v].w + e.w);
                if (ok) reinterpret_cast<float4*>(a.sum + r * a.D)[c] = xv[v];
            }
            s += (xv[v].x + xv[v].y) + (xv[v].z + xv[v].w);
        }
        const float mean = group_sum(s, G) * invD;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = li + v * G;
            if (c < D4) {
                const float dx = xv[v].x - mean, dy = xv[v].y - mean, dz = xv[v].z - mean, dw = xv[v].w - mean;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        const float rstd = rsqrtf(group_sum(q, G) * invD + a.eps);
        if (ok) {
            float4* yr = reinterpret_cast<float4*>(a.y + r * a.D);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c = li + v * G;
                if (c < D4)
                    yr[c] = make_float4((xv[v].x - mean) * rstd * gm[v].x + bt[v].x, (xv[v].y - mean) * rstd * gm[v].y + bt[v].y,
                                        (xv[v].z - mean) * rstd * gm[v].z + bt[v].z, (xv[v].w - mean) * rstd * gm[v].w + bt[v].w);
            }
            if (li == 0) { a.mean[r] = mean; a.rstd[r] = rstd; }
        }
    }
}

template <int NV>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const LnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float red[];     // [slots][2][D], slots = 4 waves * rows per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int G = a.G, rpw = 64 / G, sub = lane / G, li = lane - sub * G;
    const int D4 = a.D >> 2;
    const float invD = 1.f / (float)a.D;
    float4 gm[NV], dg[NV], db[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = li + v * G;
        gm[v] = c < D4 ? reinterpret_cast<const float4*>(a.gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[v] = db[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * a.iters * rpw;
    float am = 0.f;
    for (int it = 0; it < a.iters; ++it) {
        const long long r = row0 + (long long)it * rpw + sub;
        if (row0 + (long long)it * rpw >= a.R) break;               // wave-uniform
        const bool ok = r < a.R;
        const long long rc = ok ? r : a.R - 1;
        const float4* xr = reinterpret_cast<const float4*>(a.x + rc * a.D);
        const float4* gr = reinterpret_cast<const float4*>(a.gy + rc * a.D);
        const float mean = a.mean_in[rc], rstd = a.rstd_in[rc];
        float4 xh[NV], gh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = li + v * G;
            if (c < D4 && ok) {
                const float4 x = xr[c], g = gr[c];
                xh[v] = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
                dg[v].x += g.x * xh[v].x; dg[v].y += g.y * xh[v].y; dg[v].z += g.z * xh[v].z; dg[v].w += g.w * xh[v].w;
                db[v].x += g.x; db[v].y += g.y; db[v].z += g.z; db[v].w += g.w;
                gh[v] = make_float4(g.x * gm[v].x, g.y * gm[v].y, g.z * gm[v].z, g.w * gm[v].w);
                s1 += (gh[v].x + gh[v].y) + (gh[v].z + gh[v].w);
                s2 += (gh[v].x * xh[v].x + gh[v].y * xh[v].y) + (gh[v].z * xh[v].z + gh[v].w * xh[v].w);
            } else {
                xh[v] = gh[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float c1 = group_sum(s1, G) * invD, c2 = group_sum(s2, G) * invD;
        if (ok) {
            float4* or_ = reinterpret_cast<float4*>(a.gx + r * a.D);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int c = li + v * G;
                if (c < D4) {
                    const float4 o = make_float4((gh[v].x - c1 - xh[v].x * c2) * rstd, (gh[v].y - c1 - xh[v].y * c2) * rstd,
                                                 (gh[v].z - c1 - xh[v].z * c2) * rstd, (gh[v].w - c1 - xh[v].w * c2) * rstd);
                    or_[c] = o;
                    am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                }
            }
        }
    }
    if (a.amax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
        if (lane == 0) ign_atomic_absmax(a.amax, am);
    }
    // fold the per-lane partials of the 4 * rpw row slots of this block (fixed order), then one partial row per block
    const int slot = wave * rpw + sub, nslots = 4 * rpw;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = li + v * G;
        if (c < D4) {
            reinterpret_cast<float4*>(red + (size_t)(slot * 2 + 0) * a.D)[c] = dg[v];
            reinterpret_cast<float4*>(red + (size_t)(slot * 2 + 1) * a.D)[c] = db[v];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.D; i += 256) {
        const int which = i / a.D, col = i - which * a.D;
        float t = 0.f;
        for (int sl = 0; sl < nslots; ++sl) t += red[(size_t)(sl * 2 + which) * a.D + col];
        a.part[((size_t)blockIdx.x * 2 + which) * a.D + col] = t;
    }
}

// dgamma / dbeta[col] = sum over blocks of part[block][which][col]: 64 columns x 16 block-slices per workgroup (4 slices left each
// thread ~530 dependent-latency loads at 2000 partial rows: 33 us per call, 40 % of the backward at 25 600 rows), every slice summed
// in ascending order with 8 loads in flight, the slice sums combined in a fixed order (bitwise reproducible).  One thread per
// column walking ALL blocks (the first version) took 158 us at 2000 blocks: a serial chain of dependent loads.
constexpr int LNR_SL = 16;                             // block-slices per workgroup: 64 columns x 16 slices = 1024 threads
__global__ void __launch_bounds__(64 * LNR_SL) layernorm_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                                       float* __restrict__ dbeta, int nblk, int D) {
    __shared__ float sm[LNR_SL][64];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + o;                    // index into (which, col), 2*D in total
    const int per = (nblk + LNR_SL - 1) / LNR_SL;
    const int b0 = sl * per, b1 = min(nblk, b0 + per);
    float s = 0.f;
    if (i < 2 * D) {
        const int which = i / D, col = i - which * D;
        const float* p = part + (size_t)which * D + col;
        const size_t pitch = (size_t)2 * D;
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(b + u) * pitch];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < b1; ++b) s += p[(size_t)b * pitch];
    }
    sm[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < 2 * D) {
        const int which = i / D, col = i - which * D;
        float* out = which == 0 ? dgamma : dbeta;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < LNR_SL; ++q) t += sm[q][o];   // ascending slice order: fixed, reproducible
        if (out) out[col] = t;
    }
}

static int ln_geometry(const char* who, long long R, int D, int* G, int* nv, long long* nblk, int* iters) {
    if (R <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * LN_MAXV) {
        ign_set_error("%s: needs R > 0, D %% 4 == 0, D <= %d (R=%lld D=%d)", who, 64 * 4 * LN_MAXV, R, D);
        return IGN_E_ARG;
    }
    int g = 1;
    while (g < 64 && g < D / 4) g <<= 1;
    *G = g;
    *nv = (D / 4 + g - 1) / g;
    // 32 row groups per wave amortise the gamma / beta loads and keep the d(gamma) partials few -- but 25 600 rows of 512 (the
    // EEG-CNN encoder) would then be 200 blocks on 256 CUs (55 / 95 us per call, torch 34 / 97): aim for >= 2048 blocks
    const long long groups = (R + (64 / g) - 1) / (64 / g);              // wave-iterations in total
    int it = (int)std::min<long long>(LN_ITERS_MAX, std::max<long long>(1, groups / (4 * 2048)));
    *iters = it;
    const long long rows_per_block = 4LL * it * (64 / g);
    *nblk = (R + rows_per_block - 1) / rows_per_block;
    return 0;
}

extern "C" long long ign_layernorm_parts(long long R, int D) {
    int G, nv, iters; long long nblk;
    if (ln_geometry("ign_layernorm_parts", R, D, &G, &nv, &nblk, &iters)) return 0;
    return nblk;
}

#define IGN_LN_DISPATCH(KERNEL, nv_, grid, lds, s, args)                                                                  \
    switch (nv_) {                                                                                                        \
        case 1: hipLaunchKernelGGL((KERNEL<1>), grid, dim3(256), lds, s, args); break;                                    \
        case 2: hipLaunchKernelGGL((KERNEL<2>), grid, dim3(256), lds, s, args); break;                                    \
        case 3: case 4: hipLaunchKernelGGL((KERNEL<4>), grid, dim3(256), lds, s, args); break;                            \
        case 5: case 6: case 7: case 8: hipLaunchKernelGGL((KERNEL<8>), grid, dim3(256), lds, s, args); break;            \
        default: hipLaunchKernelGGL((KERNEL<16>), grid, dim3(256), lds, s, args); break;                                  \
    }

static int layernorm_fwd_impl(const char* who, const float* x, const float* res, float* sum, const float* gamma, const float* beta,
                              float* y, float* mean, float* rstd, long long R, int D, float eps, void* stream) {
    if (!x || !gamma || !y || !mean || !rstd) { ign_set_error("%s: null pointer", who); return IGN_E_ARG; }
    LnArgs a = {};
    a.res = res; a.sum = sum;
    long long nblk;
    int rc;
    if ((rc = ln_geometry(who, R, D, &a.G, &a.nv, &nblk, &a.iters))) return rc;
    a.x = x; a.gamma = gamma; a.beta = beta; a.y = y; a.mean = mean; a.rstd = rstd; a.R = R; a.D = D; a.eps = eps;
    IgnScopedTimer tm("layernorm_fwd", (hipStream_t)stream);
    IGN_LN_DISPATCH(layernorm_fwd_kernel, a.nv, dim3((unsigned)nblk), 0, (hipStream_t)stream, a);
    return ign_check_launch("layernorm_fwd_kernel");
}

extern "C" int ign_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                 long long R, int D, float eps, void* stream) {
    return layernorm_fwd_impl("ign_layernorm_fwd", x, nullptr, nullptr, gamma, beta, y, mean, rstd, R, D, eps, stream);
}

// y = LayerNorm(x + res): the residual connection in front of a post-norm encoder's LayerNorm inside the same pass (4 memory
// passes instead of the 5 of an add kernel + ign_layernorm_fwd); `sum` (R, D) receives x + res, which is the `x` the backward
// (ign_layernorm_bwd on `sum`) needs -- its gx is the gradient of BOTH addends.
extern "C" int ign_layernorm_res_fwd(const float* x, const float* res, float* sum, const float* gamma, const float* beta, float* y,
                                     float* mean, float* rstd, long long R, int D, float eps, void* stream) {
    if (!res || !sum) { ign_set_error("ign_layernorm_res_fwd: null res / sum"); return IGN_E_ARG; }
    return layernorm_fwd_impl("ign_layernorm_res_fwd", x, res, sum, gamma, beta, y, mean, rstd, R, D, eps, stream);
}

// part: ign_layernorm_parts(R, D) * 2 * D floats of workspace; dgamma / dbeta may be NULL
static int layernorm_bwd_impl(const char* who, const float* x, const float* gy, const float* gamma, const float* mean, const float* rstd,
                              float* gx, float* dgamma, float* dbeta, float* part, float* amax, long long R, int D, void* stream) {
    if (!x || !gy || !gamma || !mean || !rstd || !gx || !part) { ign_set_error("%s: null pointer", who); return IGN_E_ARG; }
    LnArgs a = {};
    a.amax = amax;
    long long nblk;
    int rc;
    if ((rc = ln_geometry(who, R, D, &a.G, &a.nv, &nblk, &a.iters))) return rc;
    a.x = x; a.gy = gy; a.gamma = gamma; a.mean_in = mean; a.rstd_in = rstd; a.gx = gx; a.part = part; a.R = R; a.D = D;
    const size_t lds = (size_t)4 * (64 / a.G) * 2 * D * sizeof(float);
    if (lds > 64 * 1024) { ign_set_error("%s: D=%d needs %zu bytes of LDS", who, D, lds); return IGN_E_TOOBIG; }
    hipStream_t s = (hipStream_t)stream;
    {
        IgnScopedTimer tm("layernorm_bwd", s);
        IGN_LN_DISPATCH(layernorm_bwd_kernel, a.nv, dim3((unsigned)nblk), lds, s, a);
    }
    if ((rc = ign_check_launch("layernorm_bwd_kernel"))) return rc;
    if (dgamma || dbeta) {
        hipLaunchKernelGGL(layernorm_reduce_kernel, dim3((unsigned)((2 * D + 63) / 64)), dim3(64 * LNR_SL), 0, s, part, dgamma, dbeta,
                           (int)nblk, D);
        return ign_check_launch("layernorm_reduce_kernel");
    }
    return 0;
}

extern "C" int ign_layernorm_bwd(const float* x, const float* gy, const float* gamma, const float* mean, const float* rstd,
                                 float* gx, float* dgamma, float* dbeta, float* part, long long R, int D, void* stream) {
    return layernorm_bwd_impl("ign_layernorm_bwd", x, gy, gamma, mean, rstd, gx, dgamma, dbeta, part, nullptr, R, D, stream);
}

// ... and max |gx| as an atomic maximum into *amax_slot (caller zeroes): the bound the fp16 GEMM behind it scales dL/dy by
extern "C" int ign_layernorm_bwd_amax(const float* x, const float* gy, const float* gamma, const float* mean, const float* rstd,
                                      float* gx, float* dgamma, float* dbeta, float* part, float* amax_slot, long long R, int D,
                                      void* stream) {
    if (!amax_slot) { ign_set_error("ign_layernorm_bwd_amax: null amax_slot"); return IGN_E_ARG; }
    return layernorm_bwd_impl("ign_layernorm_bwd_amax", x, gy, gamma, mean, rstd, gx, dgamma, dbeta, part, amax_slot, R, D, stream);
}

// Backward instantiations + the fixed-order reduction over batch slices.
#include "ign_shapelet_bwd.h"

shp_bwd_launch_t ign_get_bwd_launcher(int dist, int JJ) {
    if (dist == DIST_L1) {
        if (JJ == 4) return shp_bwd_launch<4, DIST_L1>;
        if (JJ == 8) return shp_bwd_launch<8, DIST_L1>;
    } else if (dist == DIST_MSE) {
        if (JJ == 4) return shp_bwd_launch<4, DIST_MSE>;
        if (JJ == 8) return shp_bwd_launch<8, DIST_MSE>;
    } else if (dist == DIST_COS) {
        if (JJ == 4) return shp_bwd_launch<4, DIST_COS>;
        if (JJ == 8) return shp_bwd_launch<8, DIST_COS>;
    } else if (dist == DIST_PEARSON) {
        if (JJ == 4) return shp_bwd_launch<4, DIST_PEARSON>;
        if (JJ == 8) return shp_bwd_launch<8, DIST_PEARSON>;
    }
    return nullptr;
}

shp_bwd_launch_t ign_get_bwd_strided_launcher(int dist) {
    switch (dist) {
        case DIST_L1: return shp_bwd_strided_launch<DIST_L1>;
        case DIST_MSE: return shp_bwd_strided_launch<DIST_MSE>;
        case DIST_COS: return shp_bwd_strided_launch<DIST_COS>;
        case DIST_PEARSON: return shp_bwd_strided_launch<DIST_PEARSON>;
    }
    return nullptr;
}

// out[i] = sum_{s < nparts} part[s][i], s ascending: bitwise reproducible.
__global__ void __launch_bounds__(256) reduce_parts_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           int nparts, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    out[i] = s;
}

// Many partials, few outputs (EEG-CNN statistics: 4096 partials of 8..1000 values): 16 outputs x 16 slices of the
// partials per block; each slice is summed in ascending order and the 16 slice sums are combined in fixed order.
__global__ void __launch_bounds__(256) reduce_parts_sliced_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                  int nparts, size_t n) {
    __shared__ float sm[16][17];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const size_t i = (size_t)blockIdx.x * 16 + o;
    const int per = (nparts + 15) / 16;
    const int p0 = sl * per, p1 = min(nparts, p0 + per);
    float s = 0.f;
    if (i < n)
        for (int p = p0; p < p1; ++p) s += part[(size_t)p * n + i];
    sm[sl][o] = s;
    __syncthreads();
    if (threadIdx.x < 16 && (size_t)blockIdx.x * 16 + threadIdx.x < n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sm[q][threadIdx.x];
        out[(size_t)blockIdx.x * 16 + threadIdx.x] = t;
    }
}

// Mid-size case (16 <= nparts <= 128, the shapelet backward's batch slices): 64 outputs x 4 slices of the partials per
// block; each slice is summed in ascending order with 8 loads in flight, the 4 slice sums are combined in fixed order.
__global__ void __launch_bounds__(256) reduce_parts_x4_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int nparts, size_t n) {
    __shared__ float sm[4][64];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + o;
    const int per = (nparts + 3) / 4;
    const int p0 = sl * per, p1 = min(nparts, p0 + per);
    float s = 0.f;
    if (i < n) {
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(p + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; p < p1; ++p) s += part[(size_t)p * n + i];
    }
    sm[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < n) out[i] = ((sm[0][o] + sm[1][o]) + sm[2][o]) + sm[3][o];
}

void ign_launch_reduce_parts(const float* part, float* out, int nparts, size_t n, hipStream_t s) {
    if (nparts > 128) {
        hipLaunchKernelGGL(reduce_parts_sliced_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, part, out, nparts, n);
        return;
    }
    if (nparts >= 16) {
        hipLaunchKernelGGL(reduce_parts_x4_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, part, out, nparts, n);
        return;
    }
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(blocks), dim3(256), 0, s, part, out, nparts, n);
}

// All groups of a bank in ONE reduction launch (blockIdx.y = group): out_g[i] = sum_s part_g[s][i] (+ scale[0] * add_g[i]).
// Same arithmetic as the per-group kernels above for up to 128 partials -- >= 16: 4 slices, each summed in ascending order, combined
// in a fixed order; fewer: one ascending sum -- so a bank call returns bitwise what G ign_shapelet_bwd calls return (nothing added).
// `add` carries a batch-independent gradient of the same tensor (the diversity regulariser, IGN/model/Shapelet.py:223-230) so
// that it needs no accumulate kernel; `scale` is the upstream gradient of that regulariser, read on the device.
__global__ void __launch_bounds__(256) reduce_bank_kernel(const ReduceBankTable t, const float* __restrict__ scale) {
    __shared__ float sm[4][64];
    const int g = blockIdx.y;
    const size_t n = t.n[g];
    const int nparts = t.nparts[g];
    const float* __restrict__ part = t.part[g];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + o;
    if ((size_t)blockIdx.x * 64 >= n) return;                       // block-uniform
    float s = 0.f;
    if (nparts >= 16) {
        const int per = (nparts + 3) / 4;
        const int p0 = sl * per, p1 = min(nparts, p0 + per);
        if (i < n) {
            int p = p0;
            for (; p + 8 <= p1; p += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(p + u) * n + i];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; p < p1; ++p) s += part[(size_t)p * n + i];
        }
        sm[sl][o] = s;
        __syncthreads();
        if (sl != 0 || i >= n) return;
        s = ((sm[0][o] + sm[1][o]) + sm[2][o]) + sm[3][o];
    } else {
        if (sl != 0 || i >= n) return;
        for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
    }
    if (t.add[g]) s += (scale ? scale[0] : 1.f) * t.add[g][i];
    t.out[g][i] = s;
}

void ign_launch_reduce_bank(const ReduceBankTable& t, int G, const float* scale_dev, hipStream_t s) {
    size_t nmax = 0;
    for (int g = 0; g < G; ++g) nmax = t.n[g] > nmax ? t.n[g] : nmax;
    hipLaunchKernelGGL(reduce_bank_kernel, dim3((unsigned)((nmax + 63) / 64), (unsigned)G), dim3(256), 0, s, t, scale_dev);
}

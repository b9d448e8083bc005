// Backward instantiations + the fixed-order reduction over batch slices.
#include "ign_shapelet_bwd.h"

shp_bwd_launch_t ign_get_bwd_launcher(int dist, int JJ) {
    if (dist == DIST_L1) {
        if (JJ == 4) return shp_bwd_launch<4, DIST_L1>;
        if (JJ == 8) return shp_bwd_launch<8, DIST_L1>;
    } else if (dist == DIST_MSE) {
        if (JJ == 4) return shp_bwd_launch<4, DIST_MSE>;
        if (JJ == 8) return shp_bwd_launch<8, DIST_MSE>;
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

void ign_launch_reduce_parts(const float* part, float* out, int nparts, size_t n, hipStream_t s) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(blocks), dim3(256), 0, s, part, out, nparts, n);
}

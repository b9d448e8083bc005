// Shared declarations of libign_hip.so (gfx950 only; no portability layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/ign_abi.h"

// max of non-negative floats as unsigned integers on their bit patterns: exact, order-independent (bitwise reproducible)
__device__ __forceinline__ void ign_atomic_absmax(float* slot, float v) {
    unsigned int* u = reinterpret_cast<unsigned int*>(slot);
    const unsigned int b = __float_as_uint(v);
    if (b > __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(u, b);
}


#define IGN_WAVE 64

// thread-local error message (ign_abi.cpp owns the storage)
void ign_set_error(const char* fmt, ...);
int  ign_check_launch(const char* what);

// RAII bracket used around kernel launches when ign_timing_enable(1) is active (ign_abi.hip)
struct IgnScopedTimer {
    IgnScopedTimer(const char* label, hipStream_t s);
    ~IgnScopedTimer();
    const char* label; hipStream_t stream; hipEvent_t e0, e1; bool on;
};

enum { DIST_L1 = 0, DIST_MSE = 1, DIST_COS = 2, DIST_PEARSON = 3 };
enum { GATE_RBF = 0, GATE_LTS = 1 };

// ---------------------------------------------------------------- forward
struct ShpFwdArgs {
    const float* xn;      // (B,C,T)
    const float* w;       // (K,C,L)
    const float* thr;     // (K,C) or null
    float* p_out;         // (B,ld) + col0
    float* dmin_out;      // (B,ld) + col0
    int32_t* tstar;       // (B,K,C)
    float* zmu;           // (B,K,C,2)
    float* d;             // (B,C,K,Tw) or null
    float* xstat;         // (B,C,Tw) or null: ||x_win|| (cosine) / sqrt(sum (x_win - mean)^2) (pearson), saved for the backward
    int B, C, T, K, L, Tw, stride, ld, col0;
    int k0;               // first shapelet of this launch (blockIdx.y adds KT each)
    int npass;            // passes of 64*TT windows per row
    int xs_len;           // floats of LDS per wave (multiple of 4)
    int gate;             // GATE_RBF / GATE_LTS
    float eps, invL;
};

constexpr int SHP_MAX_GROUPS = 8;      // groups per ign_shapelet_fwd_bank call

// launchers generated per (DIST, TT, KT); defined in ign_shapelet_fwd_*.hip
typedef void (*shp_fwd_launch_t)(const ShpFwdArgs&, dim3 grid, dim3 block, size_t lds, hipStream_t);
shp_fwd_launch_t ign_get_fwd_launcher(int dist, int TT, int KT);   // null if not instantiated

// ---------------------------------------------------------------- backward
struct ShpBwdArgs {
    const float* xn;      // (B,C,T)
    const float* w;       // (K,C,L)
    const float* g;       // (B,ld)+col0 upstream grad of p_out
    const float* p;       // (B,ld)+col0 forward gate output (LTS)
    const float* dmin;    // (B,ld)+col0 (LTS)
    const int32_t* tstar; // (B,K,C)
    const float* zmu;     // (B,K,C,2)
    const float* d;       // (B,C,K,Tw)
    const float* xstat;   // (B,C,Tw)  window norms (cosine / pearson)
    const float* wnorm;   // (K,C)     shapelet norms sqrt(sum_j w^2) (cosine / pearson)
    float* part;          // (nbs,K,C,L) partial sums over batch slices
    int B, C, T, K, L, Tw, ld, col0;
    int nbs;              // batch slices (gridDim.y)
    int kb;               // shapelets per block (gridDim.z tiles)
    int cpk;              // j-chunks per shapelet = ceil(L/JJ)
    int tc;               // window positions staged per LDS chunk (multiple of 2*JJ)
    int xs_len;           // floats of x staging per chunk = cpk*JJ + tc (multiple of 4); strided: cpk*JJ + (tc-1)*stride + 1
    int gate;
    float eps, invL;
    int stride;           // window step (1 below seq_len 3000; int(log2 L) above: IGN/model/Shapelet.py:162)
    int njt;              // strided kernel only: tiles of cpk*JJ shapelet positions per shapelet (L > 2048 needs > 1)
};
typedef void (*shp_bwd_launch_t)(const ShpBwdArgs&, dim3 grid, dim3 block, size_t lds, hipStream_t);
shp_bwd_launch_t ign_get_bwd_launcher(int dist, int JJ);            // JJ in {4,8}
shp_bwd_launch_t ign_get_bwd_strided_launcher(int dist);            // stride > 1: JJ = 4, generic window step

void ign_launch_reduce_parts(const float* part, float* out, int nparts, size_t n, hipStream_t s);

// the reductions of every group of a bank in one launch (ign_shapelet_bwd.hip)
struct ReduceBankTable {
    const float* part[SHP_MAX_GROUPS];
    const float* add[SHP_MAX_GROUPS];      // nullable: out += scale[0] * add
    float* out[SHP_MAX_GROUPS];
    size_t n[SHP_MAX_GROUPS];
    int nparts[SHP_MAX_GROUPS];
};
void ign_launch_reduce_bank(const ReduceBankTable& t, int G, const float* scale_dev, hipStream_t s);

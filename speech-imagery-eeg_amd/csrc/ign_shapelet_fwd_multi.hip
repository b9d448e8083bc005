// Shapelet forward of a whole BANK in one launch.
//
// The driver-default IGN bank has four length groups (L = 100 / 200 / 300 / 500 at T = 1000) and every group used to be its
// own launch of B*C one-wave blocks: 31 232 blocks on 4096 wave slots = 7.6 scheduling rounds, i.e. each launch ends with a
// round that is 60 % full (~5 % of its time), four times per step.  Here the blocks of all groups form ONE grid, ordered by
// work per block (longest shapelets first), so the only tail left is one round of the cheapest blocks.  The per-block code is
// the same shp_fwd_body<TT, 5, L1> as the single-group kernel (bitwise identical results); the block picks its group from a
// prefix table in the kernel arguments and its windows-per-lane count TT by a wave-uniform switch.
#include "ign_shapelet_fwd.h"

#define IGN_MULTI_CASE(TTV) case TTV: shp_fwd_body<TTV, 5, DIST_L1>(a, bx, by, smem); break;

// The group table is indexed with a run-time (wave-uniform) index.  Indexing the by-value kernel argument itself makes clang keep
// a private copy of the whole 1.2 KB struct in scratch (1.5 KB / lane); reading it through the kernarg segment pointer in the
// constant address space turns every field into an s_load with a uniform offset.
typedef const __attribute__((address_space(4))) ShpFwdMulti* cmulti_p;

__global__ void __launch_bounds__(256, 4) shp_fwd_multi_kernel(const ShpFwdMulti /* read via the kernarg pointer */) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const cmulti_p m = (cmulti_p)__builtin_amdgcn_kernarg_segment_ptr();
    const int bid = blockIdx.x;
    int gi = 0;
#pragma unroll
    for (int q = 1; q < SHP_MAX_GROUPS; ++q)
        if (q < m->ng && bid >= m->start[q]) gi = q;
    const int local = bid - m->start[gi];
    const int gx = m->gridx[gi];
    const int by = local / gx;
    const int bx = local - by * gx;
    ShpFwdArgs a;
    a.xn = m->g[gi].xn; a.w = m->g[gi].w; a.thr = m->g[gi].thr; a.p_out = m->g[gi].p_out; a.dmin_out = m->g[gi].dmin_out;
    a.tstar = m->g[gi].tstar; a.zmu = m->g[gi].zmu; a.d = m->g[gi].d; a.xstat = m->g[gi].xstat;
    a.B = m->g[gi].B; a.C = m->g[gi].C; a.T = m->g[gi].T; a.K = m->g[gi].K; a.L = m->g[gi].L; a.Tw = m->g[gi].Tw;
    a.stride = m->g[gi].stride; a.ld = m->g[gi].ld; a.col0 = m->g[gi].col0; a.k0 = m->g[gi].k0; a.npass = m->g[gi].npass;
    a.xs_len = m->g[gi].xs_len; a.gate = m->g[gi].gate; a.eps = m->g[gi].eps; a.invL = m->g[gi].invL;
    switch (m->tt[gi]) {
        IGN_MULTI_CASE(1) IGN_MULTI_CASE(2) IGN_MULTI_CASE(3) IGN_MULTI_CASE(4)
        IGN_MULTI_CASE(5) IGN_MULTI_CASE(6) IGN_MULTI_CASE(7) IGN_MULTI_CASE(8)
        IGN_MULTI_CASE(9) IGN_MULTI_CASE(10) IGN_MULTI_CASE(11) IGN_MULTI_CASE(12)
        IGN_MULTI_CASE(13) IGN_MULTI_CASE(14) IGN_MULTI_CASE(15) IGN_MULTI_CASE(16)
        default: break;
    }
}

void ign_launch_shp_fwd_multi(const ShpFwdMulti& m, int nblocks, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL(shp_fwd_multi_kernel, dim3((unsigned)nblocks), dim3(64), lds, s, m);
}

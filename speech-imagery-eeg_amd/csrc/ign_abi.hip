// C-ABI entry points of libign_hip.so: argument validation and launch planning for the shapelet kernels.
#include "ign_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <algorithm>

static thread_local char g_err[512] = "";

void ign_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ign_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ign_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ timing
#include <map>
#include <mutex>
#include <string>
#include <vector>
namespace {
struct TimingSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0.0;
    long long launches = 0;
};
bool g_timing = false;
std::mutex g_tm;
std::map<std::string, TimingSlot> g_slots;
void drain(TimingSlot& sl) {
    for (auto& pr : sl.pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            sl.total_ms += ms;
            sl.launches += 1;
        }
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    sl.pending.clear();
}
}  // namespace

IgnScopedTimer::IgnScopedTimer(const char* l, hipStream_t s) : label(l), stream(s), e0(nullptr), e1(nullptr), on(g_timing) {
    if (!on) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(e0, stream);
}
IgnScopedTimer::~IgnScopedTimer() {
    if (!on) return;
    (void)hipEventRecord(e1, stream);
    std::lock_guard<std::mutex> lk(g_tm);
    g_slots[label].pending.emplace_back(e0, e1);
}

extern "C" int ign_timing_enable(int on) {
    std::lock_guard<std::mutex> lk(g_tm);
    for (auto& kv : g_slots) drain(kv.second);
    g_slots.clear();
    g_timing = on != 0;
    return 0;
}

extern "C" int ign_timing_read(const char* label, double* total_ms, long long* launches) {
    if (!label || !total_ms || !launches) { ign_set_error("ign_timing_read: null argument"); return IGN_E_ARG; }
    std::lock_guard<std::mutex> lk(g_tm);
    auto it = g_slots.find(label);
    if (it == g_slots.end()) { *total_ms = 0.0; *launches = 0; return 0; }
    drain(it->second);
    *total_ms = it->second.total_ms;
    *launches = it->second.launches;
    return 0;
}

extern "C" int ign_abi_version(void) { return IGN_ABI_VERSION; }
extern "C" const char* ign_last_error(void) { return g_err; }

extern shp_fwd_launch_t ign_fwd_table_p0[4][4][3];
extern shp_fwd_launch_t ign_fwd_table_p1[4][4][3];
extern shp_fwd_launch_t ign_fwd_table_p2[4][4][3];
extern shp_fwd_launch_t ign_fwd_table_p3[4][4][3];

shp_fwd_launch_t ign_get_fwd_launcher(int dist, int TT, int KT) {
    if (dist < 0 || dist > 3 || TT < 1 || TT > 16) return nullptr;
    const int ki = KT == 1 ? 0 : KT == 2 ? 1 : KT == 5 ? 2 : -1;
    if (ki < 0) return nullptr;
    shp_fwd_launch_t (*tabs[4])[4][3] = {ign_fwd_table_p0, ign_fwd_table_p1, ign_fwd_table_p2, ign_fwd_table_p3};
    return tabs[(TT - 1) / 4][(TT - 1) % 4][dist][ki];
}

static int split_mode(int mode, int* dist, int* gate, const char* who) {
    *dist = mode & 0xf;
    *gate = (mode & IGN_GATE_LTS) ? GATE_LTS : GATE_RBF;
    if ((mode & ~0x1f) != 0 || *dist > IGN_DIST_PEARS) {
        ign_set_error("%s: unknown mode 0x%x", who, mode);
        return IGN_E_ARG;
    }
    return 0;
}

static int check_dims(const char* who, int B, int C, int T, int K, int L, int stride) {
    if (B <= 0 || C <= 0 || T <= 0 || K <= 0 || L <= 0 || stride <= 0 || L > T) {
        ign_set_error("%s: bad dimensions B=%d C=%d T=%d K=%d L=%d stride=%d", who, B, C, T, K, L, stride);
        return IGN_E_ARG;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ forward
struct FwdPlan {
    ShpFwdArgs a;
    int TT, wpb, dist;
    size_t lds;
};

static int plan_fwd(const char* who, const float* xn_bct, const float* w_kcl, const float* thr_kc, float* p_out, float* dmin_out,
                    int ld, int col0, int32_t* tstar, float* zmu, float* d_save, float* xstat_save, int B, int C, int T, int K,
                    int L, int stride, float eps, int mode, FwdPlan* pl) {
    int dist, gate, rc;
    if ((rc = split_mode(mode, &dist, &gate, who))) return rc;
    if ((rc = check_dims(who, B, C, T, K, L, stride))) return rc;
    if (!xn_bct || !w_kcl || !p_out || !dmin_out || !tstar || !zmu || (gate == GATE_LTS && !thr_kc)) {
        ign_set_error("%s: null pointer argument", who);
        return IGN_E_ARG;
    }
    if (ld < col0 + K * C || col0 < 0) {
        ign_set_error("%s: output row pitch ld=%d too small for col0=%d + K*C=%d", who, ld, col0, K * C);
        return IGN_E_ARG;
    }
    const int Tw = (T - L) / stride + 1;
    // TT windows per lane: one wave pass covers the row when Tw <= 1024; strided windows use TT = 1.
    int TT = (stride == 1) ? std::min(16, (Tw + 63) / 64) : 1;
    const int npass = (Tw + 64 * TT - 1) / (64 * TT);
    int xs_len = (npass * 64 * TT - 1) * stride + (TT - 1) + L;
    xs_len = (xs_len + 3) & ~3;
    int wpb = 1;      // one wave per block: waves share nothing, and the epilogue's __syncthreads() stays wave-local
    const size_t park = (npass > 1) ? (size_t)5 * 5 * 64 * 4 : 0;     // per wave: 5 stats x KT<=5 x 64 lanes
    const size_t lds = (size_t)wpb * (xs_len * 4 + park);
    if (lds > 160 * 1024) {       // the whole LDS of a gfx950 CU: rows up to T ~ 40 000 (the longest UEA set is 17 984)
        ign_set_error("%s: a row needs %zu bytes of LDS staging (T=%d L=%d stride=%d)", who, lds, T, L, stride);
        return IGN_E_TOOBIG;
    }
    ShpFwdArgs& a = pl->a;
    a.xn = xn_bct; a.w = w_kcl; a.thr = thr_kc; a.p_out = p_out; a.dmin_out = dmin_out; a.tstar = tstar; a.zmu = zmu;
    a.d = d_save;
    a.xstat = (dist >= DIST_COS) ? xstat_save : nullptr;
    a.B = B; a.C = C; a.T = T; a.K = K; a.L = L; a.Tw = Tw; a.stride = stride; a.ld = ld; a.col0 = col0;
    a.npass = npass; a.xs_len = xs_len; a.gate = gate; a.eps = eps; a.invL = 1.0f / (float)L;
    a.k0 = 0;
    pl->TT = TT; pl->wpb = wpb; pl->dist = dist; pl->lds = lds;
    return 0;
}

static int launch_fwd_group(const char* who, FwdPlan& pl, void* stream) {
    ShpFwdArgs& a = pl.a;
    const int nbg = (a.B + pl.wpb - 1) / pl.wpb;
    // shapelets in tiles of 5, then 2, then 1 (each tile shares the x registers across its shapelets)
    int k0 = 0, rc;
    static const int tiles[3] = {5, 2, 1};
    for (int ti = 0; ti < 3; ++ti) {
        const int KT = tiles[ti];
        const int n = (a.K - k0) / KT;
        if (n <= 0) continue;
        shp_fwd_launch_t fn = ign_get_fwd_launcher(pl.dist, pl.TT, KT);
        if (!fn) {
            ign_set_error("%s: no kernel for TT=%d KT=%d", who, pl.TT, KT);
            return IGN_E_UNSUP;
        }
        a.k0 = k0;
        {
            IgnScopedTimer tm("shp_fwd", (hipStream_t)stream);
            fn(a, dim3((unsigned)a.C * nbg, (unsigned)n), dim3(pl.wpb * 64), pl.lds, (hipStream_t)stream);
        }
        if ((rc = ign_check_launch("shp_fwd_kernel"))) return rc;
        k0 += n * KT;
    }
    return 0;
}

extern "C" int ign_shapelet_fwd(const float* xn_bct, const float* w_kcl, const float* thr_kc, float* p_out,
                                float* dmin_out, int ld, int col0, int32_t* tstar, float* zmu, float* d_save,
                                float* xstat_save, int B, int C, int T, int K, int L, int stride, float eps, int mode,
                                void* stream) {
    static const char* who = "ign_shapelet_fwd";
    FwdPlan pl;
    int rc;
    if ((rc = plan_fwd(who, xn_bct, w_kcl, thr_kc, p_out, dmin_out, ld, col0, tstar, zmu, d_save, xstat_save, B, C, T, K, L, stride,
                       eps, mode, &pl))) return rc;
    return launch_fwd_group(who, pl, stream);
}

// All G length groups of a bank (IGN/model/Shapelet.py:190-196: the loop over self.shapelets) in one call, launched group by
// group.  A single grid over the blocks of all groups (heaviest first, one scheduling tail per step instead of G) was built and
// measured in round 2 (shp_fwd_multi_kernel: the same shp_fwd_body behind a wave-uniform switch over TT): 4.07 ms against 3.97 ms
// for the four separate launches at the benchmark shape on the same box -- the merged kernel carries the largest body's scratch
// and register allocation into every block -- so it was removed again (tests/diag_fwd_bank.py keeps the A/B harness).
extern "C" int ign_shapelet_fwd_bank(const float* xn_bct, int G, const float* const* w_kcl, const float* const* thr_kc, float* p_out,
                                     float* dmin_out, int ld, const int* col0, int32_t* const* tstar, float* const* zmu,
                                     float* const* d_save, float* const* xstat_save, int B, int C, int T, const int* K, const int* L,
                                     const int* stride, float eps, int mode, void* stream) {
    static const char* who = "ign_shapelet_fwd_bank";
    if (G <= 0 || G > SHP_MAX_GROUPS || !w_kcl || !col0 || !tstar || !zmu || !K || !L || !stride) {
        ign_set_error("%s: G=%d outside 1..%d or null table", who, G, SHP_MAX_GROUPS);
        return IGN_E_ARG;
    }
    FwdPlan pl[SHP_MAX_GROUPS];
    int rc;
    for (int g = 0; g < G; ++g)          // validate every group before the first launch
        if ((rc = plan_fwd(who, xn_bct, w_kcl[g], thr_kc ? thr_kc[g] : nullptr, p_out, dmin_out, ld, col0[g], tstar[g], zmu[g],
                           d_save ? d_save[g] : nullptr, xstat_save ? xstat_save[g] : nullptr, B, C, T, K[g], L[g], stride[g], eps,
                           mode, &pl[g]))) return rc;
    // Launch order: longest shapelets first.  The groups are independent (results do not depend on the order), but the tail of a
    // group's grid -- its last, partly filled round of blocks -- is covered by the start of the next group's grid, so only the LAST
    // group's tail is exposed, and the group with the shortest blocks has the shortest one (same box, four orders x 60 runs:
    // 3.85-3.86 ms ending on L = 100 or 300 against 3.91 ms ending on L = 500).
    int order[SHP_MAX_GROUPS];
    for (int g = 0; g < G; ++g) order[g] = g;
#ifndef IGN_FWD_GROUP_ORDER_AS_GIVEN
    for (int i = 1; i < G; ++i) {                 // insertion sort, descending L, stable
        const int v = order[i];
        int j = i - 1;
        while (j >= 0 && L[order[j]] < L[v]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = v;
    }
#endif
    for (int i = 0; i < G; ++i)
        if ((rc = launch_fwd_group(who, pl[order[i]], stream))) return rc;
    return 0;
}

// ------------------------------------------------------------------------------------------ backward
struct BwdPlan {
    int JJ, cpk, kb, nkt, threads, tc, xs_len, nbs, njt;
    size_t lds;
};

// stride > 1 (IGN/model/Shapelet.py:162 at seq_len >= 3000): the generic-step kernel, JJ = 4; one shapelet per block when
// it needs more than one block of 512 lanes (L > 2048), otherwise the lane-filling tile of shapelets as below.
static int plan_bwd_strided(int B, int C, int K, int L, int Tw, int stride, BwdPlan* p) {
    const int JJ = 4;
    p->JJ = JJ;
    const int cpk_all = (L + JJ - 1) / JJ;
    p->njt = (cpk_all + 511) / 512;
    p->cpk = (cpk_all + p->njt - 1) / p->njt;            // j-chunks per tile, equalised over the tiles
    if (p->njt > 1) {
        p->kb = 1;
    } else {
        int best_kb = 1;
        double best_u = -1.0;
        for (int kb = 1; kb <= K && kb * p->cpk <= 512; ++kb) {
            const int thr = ((kb * p->cpk + 63) / 64) * 64;
            const int ntile = (K + kb - 1) / kb;
            const double u = (double)K * p->cpk / ((double)ntile * thr);
            if (u > best_u + 1e-9) { best_u = u; best_kb = kb; }
        }
        p->kb = best_kb;
    }
    p->nkt = (K + p->kb - 1) / p->kb;
    p->threads = ((p->kb * p->cpk + 63) / 64) * 64;
    const size_t budget = std::max<size_t>(16 * 1024, (size_t)(p->threads / 64) * 5 * 1024);
    const long fixed = (long)p->cpk * JJ + 16 * p->kb + 8;
    long tc_max = ((long)(budget / 4) - fixed) / (stride + p->kb);
    tc_max = std::max<long>(2 * JJ, (tc_max / (2 * JJ)) * (2 * JJ));
    const int nchunk = (int)((Tw + tc_max - 1) / tc_max);
    int tc = (Tw + nchunk - 1) / nchunk;
    tc = ((tc + 2 * JJ - 1) / (2 * JJ)) * (2 * JJ);
    p->tc = tc;
    p->xs_len = (p->cpk * JJ + (tc - 1) * stride + 1 + 3) & ~3;
    p->lds = ((size_t)p->xs_len + (size_t)p->kb * tc + 16 * (size_t)p->kb) * 4;
    if (p->lds > 64 * 1024) return IGN_E_TOOBIG;
    int nbs = (B + 1) / 2;
    while (nbs > 1 && (size_t)nbs * K * C * L * 4 > ((size_t)256 << 20)) nbs = (nbs + 1) / 2;
    p->nbs = std::max(1, std::min(nbs, B));
    return 0;
}

static int plan_bwd(int B, int C, int T, int K, int L, int Tw, int stride, BwdPlan* p) {
    if (stride > 1) return plan_bwd_strided(B, C, K, L, Tw, stride, p);
    p->njt = 1;
    // JJ: shapelet positions per lane.  8 amortises the A / x operand reads over more work; 4 fills the waves better
    // for short shapelets (K*ceil(L/JJ) lanes are rounded up to whole waves).
    int best = 0;
    double best_eff = -1.0;
    const int cand[2] = {8, 4};
    for (int i = 0; i < 2; ++i) {
        const int JJ = cand[i];
        const int cpk = (L + JJ - 1) / JJ;
        if (cpk > 512) continue;
        const int kb = std::max(1, std::min(K, 512 / cpk));
        const int threads = ((kb * cpk + 63) / 64) * 64;
        double eff = (double)kb * L / ((double)threads * JJ);
        if (JJ == 8) eff *= 1.10;
        if (eff > best_eff) { best_eff = eff; best = JJ; }
    }
    if (!best) return IGN_E_TOOBIG;
    const int JJ = best;
    p->JJ = JJ;
    p->cpk = (L + JJ - 1) / JJ;
    // shapelets per block: the tile that wastes the fewest lanes when kb*cpk is rounded up to whole waves; on ties the
    // SMALLER block (measured: 1-wave blocks at L=500 beat 5-wave blocks, 17 -> 24 T elements/s -- less barrier skew).
    {
        int best_kb = 1;
        double best_u = -1.0;
        for (int kb = 1; kb <= K && kb * p->cpk <= 512; ++kb) {
            const int thr = ((kb * p->cpk + 63) / 64) * 64;
            const int ntile = (K + kb - 1) / kb;
            const double u = (double)K * p->cpk / ((double)ntile * thr);      // useful lanes over all tiles of this size
            if (u > best_u + 1e-9) { best_u = u; best_kb = kb; }
        }
        p->kb = best_kb;
    }
    p->nkt = (K + p->kb - 1) / p->kb;
    p->threads = ((p->kb * p->cpk + 63) / 64) * 64;
    // LDS budget ~5 KB per wave keeps 7-8 waves per SIMD resident (the v_cmpx loop is latency-bound per wave):
    // stage the window axis in chunks of tc positions, equalised over ceil(Tw / tc_max) chunks.
    const size_t budget = std::max<size_t>(8 * 1024, (size_t)(p->threads / 64) * 5 * 1024);
    const long fixed = (long)p->cpk * JJ + 16 * p->kb + 8;                      // floats besides the tc-proportional part
    long tc_max = ((long)(budget / 4) - fixed) / (1 + p->kb);
    tc_max = std::max<long>(2 * JJ, (tc_max / (2 * JJ)) * (2 * JJ));
    const int nchunk = (int)((Tw + tc_max - 1) / tc_max);
    int tc = (Tw + nchunk - 1) / nchunk;
    tc = ((tc + 2 * JJ - 1) / (2 * JJ)) * (2 * JJ);
    p->tc = tc;
    p->xs_len = (p->cpk * JJ + tc + 3) & ~3;
    p->lds = ((size_t)p->xs_len + (size_t)p->kb * tc + 16 * (size_t)p->kb) * 4;      // x chunk, A, per-wave sums, scalars
    if (p->lds > 64 * 1024) return IGN_E_TOOBIG;
    // batch slices: two rows per block.  Many small blocks keep the last scheduling round of the 256 CUs short -- with
    // ~2000 blocks of 2-5 waves a third of the launch was tail (profiles/r1a); rows per block 4 -> 2 -> 1 measured
    // 5.60 -> 5.45 -> 5.35 ms over the four groups (L=500 reaches the 24.4 T elements/s issue ceiling) against a partial
    // buffer that doubles each time.  Partials: nbs*K*C*L floats, capped at 256 MB.
    int nbs = (B + 1) / 2;
    while (nbs > 1 && (size_t)nbs * K * C * L * 4 > ((size_t)256 << 20)) nbs = (nbs + 1) / 2;
    p->nbs = std::max(1, std::min(nbs, B));
    return 0;
}

extern "C" size_t ign_shapelet_bwd_workspace_bytes(int B, int C, int T, int K, int L, int stride, int mode) {
    (void)mode;
    if (B <= 0 || C <= 0 || T <= 0 || K <= 0 || L <= 0 || stride < 1 || L > T) return 0;
    BwdPlan p;
    if (plan_bwd(B, C, T, K, L, (T - L) / stride + 1, stride, &p)) return 0;
    return (size_t)p.nbs * K * C * L * sizeof(float);
}

// validate + launch the backward kernel of one group (partials into `workspace`); *nbs_out = batch slices to reduce
static int launch_bwd_group(const char* who, const float* xn_bct, const float* w_kcl, const float* g_out, const float* p_out,
                            const float* dmin_out, int ld, int col0, const int32_t* tstar, const float* zmu,
                            const float* d_save, const float* xstat_save, const float* wnorm_kc, float* gw_kcl,
                            void* workspace, int B, int C, int T, int K, int L, int stride, float eps, int mode,
                            void* stream, int* nbs_out, bool launch) {
    int dist, gate, rc;
    if ((rc = split_mode(mode, &dist, &gate, who))) return rc;
    if ((rc = check_dims(who, B, C, T, K, L, stride))) return rc;
    if (!xn_bct || !w_kcl || !g_out || !tstar || !zmu || !d_save || !gw_kcl || !workspace ||
        (gate == GATE_LTS && (!p_out || !dmin_out))) {
        ign_set_error("%s: null pointer argument (d_save is required: run the forward with d_save)", who);
        return IGN_E_ARG;
    }
    if (ld < col0 + K * C || col0 < 0) {
        ign_set_error("%s: row pitch ld=%d too small for col0=%d + K*C=%d", who, ld, col0, K * C);
        return IGN_E_ARG;
    }
    if (dist >= DIST_COS && (!xstat_save || !wnorm_kc)) {
        ign_set_error("%s: cosine / pearson need xstat_save (from the forward) and wnorm_kc", who);
        return IGN_E_ARG;
    }
    const int Tw = (T - L) / stride + 1;
    BwdPlan p;
    if ((rc = plan_bwd(B, C, T, K, L, Tw, stride, &p))) {
        ign_set_error("%s: no launch plan for K=%d L=%d Tw=%d stride=%d", who, K, L, Tw, stride);
        return rc;
    }
    shp_bwd_launch_t fn = (stride > 1) ? ign_get_bwd_strided_launcher(dist) : ign_get_bwd_launcher(dist, p.JJ);
    if (!fn) {
        ign_set_error("%s: no kernel for JJ=%d dist=%d", who, p.JJ, dist);
        return IGN_E_UNSUP;
    }
    ShpBwdArgs a;
    a.xn = xn_bct; a.w = w_kcl; a.g = g_out; a.p = p_out; a.dmin = dmin_out; a.tstar = tstar; a.zmu = zmu; a.d = d_save;
    a.xstat = xstat_save; a.wnorm = wnorm_kc;
    a.xstat = xstat_save; a.wnorm = wnorm_kc;
    a.part = (float*)workspace;
    a.B = B; a.C = C; a.T = T; a.K = K; a.L = L; a.Tw = Tw; a.ld = ld; a.col0 = col0;
    a.nbs = p.nbs; a.kb = p.kb; a.cpk = p.cpk; a.tc = p.tc; a.xs_len = p.xs_len; a.gate = gate;
    a.eps = eps; a.invL = 1.0f / (float)L;
    a.stride = stride; a.njt = p.njt;
    *nbs_out = p.nbs;
    if (!launch) return 0;
    {
        IgnScopedTimer tm("shp_bwd", (hipStream_t)stream);
        fn(a, dim3((unsigned)C, (unsigned)p.nbs, (unsigned)(p.nkt * p.njt)), dim3(p.threads), p.lds, (hipStream_t)stream);
    }
    return ign_check_launch("shp_bwd_kernel");
}

extern "C" int ign_shapelet_bwd(const float* xn_bct, const float* w_kcl, const float* g_out, const float* p_out,
                                const float* dmin_out, int ld, int col0, const int32_t* tstar, const float* zmu,
                                const float* d_save, const float* xstat_save, const float* wnorm_kc, float* gw_kcl,
                                void* workspace, int B, int C, int T, int K, int L, int stride, float eps, int mode,
                                void* stream) {
    int rc, nbs = 0;
    if ((rc = launch_bwd_group("ign_shapelet_bwd", xn_bct, w_kcl, g_out, p_out, dmin_out, ld, col0, tstar, zmu, d_save, xstat_save,
                               wnorm_kc, gw_kcl, workspace, B, C, T, K, L, stride, eps, mode, stream, &nbs, true))) return rc;
    IgnScopedTimer tm2("reduce_parts", (hipStream_t)stream);
    ign_launch_reduce_parts((const float*)workspace, gw_kcl, nbs, (size_t)K * C * L, (hipStream_t)stream);
    return ign_check_launch("reduce_parts_kernel");
}

// Every group of a bank: G backward kernels, then ONE reduction launch that also adds `add_scale_dev[0] * gw_add[g]` (the
// batch-independent gradient of the diversity regulariser) into each group's result.  Workspace: the groups' partial buffers
// back to back, each ign_shapelet_bwd_workspace_bytes(...) rounded up to 256 bytes.
extern "C" size_t ign_shapelet_bwd_bank_workspace_bytes(int G, int B, int C, int T, const int* K, const int* L, const int* stride,
                                                        int mode) {
    if (G <= 0 || G > SHP_MAX_GROUPS || !K || !L || !stride) return 0;
    size_t tot = 0;
    for (int g = 0; g < G; ++g) {
        const size_t b = ign_shapelet_bwd_workspace_bytes(B, C, T, K[g], L[g], stride[g], mode);
        if (!b) return 0;
        tot += (b + 255) & ~(size_t)255;
    }
    return tot;
}

extern "C" int ign_shapelet_bwd_bank(const float* xn_bct, int G, const float* const* w_kcl, const float* g_out, const float* p_out,
                                     const float* dmin_out, int ld, const int* col0, const int32_t* const* tstar,
                                     const float* const* zmu, const float* const* d_save, const float* const* xstat_save,
                                     const float* const* wnorm_kc, float* const* gw_kcl, const float* const* gw_add,
                                     const float* add_scale_dev, void* workspace, int B, int C, int T, const int* K, const int* L,
                                     const int* stride, float eps, int mode, void* stream) {
    static const char* who = "ign_shapelet_bwd_bank";
    if (G <= 0 || G > SHP_MAX_GROUPS || !w_kcl || !col0 || !tstar || !zmu || !d_save || !gw_kcl || !K || !L || !stride || !workspace) {
        ign_set_error("%s: G=%d outside 1..%d or null table", who, G, SHP_MAX_GROUPS);
        return IGN_E_ARG;
    }
    ReduceBankTable t;
    char* ws[SHP_MAX_GROUPS];
    int rc;
    // (launch order as given: longest-first, which helps the forward bank, measured neutral-to-worse here)
    for (int pass = 0; pass < 2; ++pass) {            // pass 0 validates every group, pass 1 launches
        char* cur = (char*)workspace;
        for (int g = 0; g < G; ++g) {
            const size_t b = ign_shapelet_bwd_workspace_bytes(B, C, T, K[g], L[g], stride[g], mode);
            if (!b) { ign_set_error("%s: group %d: no launch plan for K=%d L=%d stride=%d", who, g, K[g], L[g], stride[g]); return IGN_E_ARG; }
            ws[g] = cur;
            cur += (b + 255) & ~(size_t)255;
            int nbs = 0;
            if ((rc = launch_bwd_group(who, xn_bct, w_kcl[g], g_out, p_out, dmin_out, ld, col0[g], tstar[g], zmu[g], d_save[g],
                                       xstat_save ? xstat_save[g] : nullptr, wnorm_kc ? wnorm_kc[g] : nullptr, gw_kcl[g], ws[g], B, C,
                                       T, K[g], L[g], stride[g], eps, mode, stream, &nbs, pass == 1))) return rc;
            t.part[g] = (const float*)ws[g];
            t.add[g] = gw_add ? gw_add[g] : nullptr;
            t.out[g] = gw_kcl[g];
            t.n[g] = (size_t)K[g] * C * L[g];
            t.nparts[g] = nbs;
        }
    }
    IgnScopedTimer tm2("reduce_parts", (hipStream_t)stream);
    ign_launch_reduce_bank(t, G, add_scale_dev, (hipStream_t)stream);
    return ign_check_launch("reduce_bank_kernel");
}

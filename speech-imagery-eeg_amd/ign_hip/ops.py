"""Autograd ops over the C ABI (include/ign_abi.h).  torch supplies device memory, streams and autograd
plumbing; the arithmetic is in libign_hip.so.  Every op raises on CPU tensors: there is no fallback path."""
import ctypes

import torch

from . import _lib

import os

DIST_L1, DIST_MSE, DIST_COS, DIST_PEARSON = 0, 1, 2, 3
ATTN_MATH = os.environ.get("IGN_ATTN_MATH", "bf16x6")          # "f32": attention core on the fp32-MFMA kernels
LAYERNORM_MIN_ROWS = 0        # round 1 kept torch below 64k rows; with the row-count-aware grid and the parallel reduce the HIP kernels win everywhere
LINEAR_WGRAD = "bf16x6"       # weight gradient of ops.linear on the split kernels; "f32" (set by tests / diag scripts): the fp32-MFMA TN kernel
GATE_RBF, GATE_LTS = 0x00, 0x10


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_stream = _lib.stream


def _need_gpu(name, *ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.IgnError(f"{name}: tensor on {t.device}; the IGN hot path runs on the MI355X only "
                                f"(no CPU fallback -- the CPU restatement lives in oracle/ for tests)")
        if t.dtype != torch.float32:
            raise _lib.IgnError(f"{name}: expected float32, got {t.dtype}")


def instance_norm(x_btc, want_raw=False, eps=1e-8, input_bound=False):
    """(B,T,C) -> normalised (B,C,T) [+ raw transpose].  Replaces IGN/model/Shapelet.py:186-187.
    `input_bound`: also attach max |x| to `x_btc` (see cached_bound) for a consumer of the raw batch."""
    _need_gpu("instance_norm", x_btc)
    x = x_btc.contiguous()
    B, T, C = x.shape
    xn = torch.empty(B, C, T, device=x.device, dtype=torch.float32)
    xt = torch.empty_like(xn) if want_raw else None
    L = _lib.lib()
    if input_bound and cached_bound(x_btc) is None:
        # the raw batch is staged in LDS by this pass anyway: take max |x| here, for the FCN expert's first fp16 GEMM
        # (ign_hip/fcn.py looks the bound up on the tensor) instead of a separate pass over x
        slot = _new_slot(x.device)
        _lib.check(L.ign_instnorm_fwd_amax(_ptr(x), _ptr(xn), _ptr(xt), B, T, C, eps, _ptr(slot), _stream()), "ign_instnorm_fwd_amax")
        set_bound(x_btc, slot)
    else:
        _lib.check(L.ign_instnorm_fwd(_ptr(x), _ptr(xn), _ptr(xt), B, T, C, eps, _stream()), "ign_instnorm_fwd")
    return xn, xt


def contiguous_bct(x):
    """A contiguous (B, C, T) tensor with the contents of `x` (B, C, T).  If `x` is the `permute(0, 2, 1)` view of a contiguous
    time-first (B, T, C) batch -- what the harness hands the EEG-CNN baseline -- the transpose runs on ign_transpose_btc_to_bct
    (no gradient: input data); anything else takes torch's `.contiguous()`."""
    if x.is_contiguous():
        return x
    B, C, T = x.shape
    if (x.is_cuda and x.dtype == torch.float32 and not x.requires_grad and x.stride() == (T * C, 1, C) and B <= 65535
            and x.data_ptr() % 16 == 0):
        out = torch.empty(B, C, T, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().ign_transpose_btc_to_bct(_ptr(x), _ptr(out), B, T, C, _stream()), "ign_transpose_btc_to_bct")
        return out
    return x.contiguous()


def standardise_nct_to_btc(x_nct, eps=1e-8):
    """Raw (B,C,T) recordings -> per-sample / per-channel standardised (B,T,C) batch on the GPU (the CHISCO loader's
    Normalizer('per_sample_std') + item transpose, IGN/data_factory/eeg.py:332-367, for the whole batch at once)."""
    _need_gpu("standardise_nct_to_btc", x_nct)
    x = x_nct.contiguous()
    B, C, T = x.shape
    out = torch.empty(B, T, C, device=x.device, dtype=torch.float32)
    ws = torch.empty(B * C * 2, device=x.device, dtype=torch.float32)
    _lib.check(_lib.lib().ign_standardise_nct_to_btc(_ptr(x), _ptr(out), _ptr(ws), B, C, T, eps, _stream()),
               "ign_standardise_nct_to_btc")
    return out


def _tables(G):
    return ctypes.c_void_p * G, ctypes.c_int * G


def _pv(ts):
    return (ctypes.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])


def _bank_forward(xn, ws, thrs, eps, mode, strides, need_grad):
    """Forward of every length group of a bank: -> (P, D, saved) with P / D (B, sum_g K_g*C) in the reference's feature order
    g*K*C + k*C + c (IGN/model/Shapelet.py:84,195-196) and saved[g] = (tstar, zmu, dsave, col0, stride, xstat)."""
    B, C, T = xn.shape
    ld = sum(w.shape[0] * C for w in ws)
    P = torch.empty(B, ld, device=xn.device, dtype=torch.float32)
    D = torch.empty_like(P)
    L = _lib.lib()
    saved, col0 = [], 0
    for g, w in enumerate(ws):
        K, Cw, Lg = w.shape
        if Cw != C:
            raise _lib.IgnError(f"shapelet group {g}: weights have {Cw} channels, input has {C}")
        stride = int(strides[g])
        Tw = (T - Lg) // stride + 1
        tstar = torch.empty(B, K, C, device=xn.device, dtype=torch.int32)
        zmu = torch.empty(B, K, C, 2, device=xn.device, dtype=torch.float32)
        dsave = torch.empty(B, C, K, Tw, device=xn.device, dtype=torch.float32) if need_grad else None
        xstat = torch.empty(B, C, Tw, device=xn.device, dtype=torch.float32) \
            if (need_grad and (mode & 0xf) >= DIST_COS) else None
        saved.append((tstar, zmu, dsave, col0, stride, xstat))
        col0 += K * C
    G = len(ws)
    if G <= 8:
        # the whole bank in one call (ign_shapelet_fwd_bank validates every group, then launches group by group)
        _, ia = _tables(G)
        _lib.check(L.ign_shapelet_fwd_bank(
            _ptr(xn), G, _pv(ws), _pv(thrs), _ptr(P), _ptr(D), ld, ia(*[sv[3] for sv in saved]), _pv([sv[0] for sv in saved]),
            _pv([sv[1] for sv in saved]), _pv([sv[2] for sv in saved]), _pv([sv[5] for sv in saved]), B, C, T,
            ia(*[w.shape[0] for w in ws]), ia(*[w.shape[2] for w in ws]), ia(*[sv[4] for sv in saved]), float(eps), int(mode),
            _stream()), "ign_shapelet_fwd_bank")
    else:
        for g, w in enumerate(ws):
            tstar, zmu, dsave, c0, stride, xstat = saved[g]
            _lib.check(L.ign_shapelet_fwd(_ptr(xn), _ptr(w), _ptr(thrs[g]), _ptr(P), _ptr(D), ld, c0, _ptr(tstar), _ptr(zmu),
                                          _ptr(dsave), _ptr(xstat), B, C, T, w.shape[0], w.shape[2], stride, float(eps), int(mode),
                                          _stream()), "ign_shapelet_fwd")
    return P, D, saved


def _cat_tstar(saved, B):
    return torch.cat([sv[0].reshape(B, -1) for sv in saved], dim=1) if len(saved) > 1 else saved[0][0].reshape(B, -1)


def _bank_backward(xn, ws, gP, P, D, saved, eps, mode, gw_add=None, add_scale=None):
    """dloss/dw of every group (list of (K,C,L) tensors) from gP = dloss/dP (B, ld).  `gw_add[g]` / `add_scale` (a one-element
    device tensor): a batch-independent gradient of the same shapelets that the reduction launch adds as add_scale * gw_add[g]
    (the diversity regulariser) -- a parameter with two gradient sources then needs no accumulate kernel."""
    B, C, T = xn.shape
    ld = P.shape[1]
    L = _lib.lib()
    G = len(ws)
    cos = (mode & 0xf) >= DIST_COS
    wnorms = [w.square().sum(dim=-1).sqrt().contiguous() if cos else None for w in ws]
    gws = [torch.empty_like(w) for w in ws]
    if G <= 8:
        _, ia = _tables(G)
        Ks, Ls, Ss = ia(*[w.shape[0] for w in ws]), ia(*[w.shape[2] for w in ws]), ia(*[sv[4] for sv in saved])
        nbytes = L.ign_shapelet_bwd_bank_workspace_bytes(G, B, C, T, Ks, Ls, Ss, mode)
        if nbytes == 0:
            raise _lib.IgnError(f"shapelet backward: no launch plan for K={list(Ks)} L={list(Ls)} stride={list(Ss)}")
        work = torch.empty(nbytes // 4, device=xn.device, dtype=torch.float32)
        _lib.check(L.ign_shapelet_bwd_bank(
            _ptr(xn), G, _pv(ws), _ptr(gP), _ptr(P), _ptr(D), ld, ia(*[sv[3] for sv in saved]), _pv([sv[0] for sv in saved]),
            _pv([sv[1] for sv in saved]), _pv([sv[2] for sv in saved]), _pv([sv[5] for sv in saved]), _pv(wnorms), _pv(gws),
            _pv(gw_add) if gw_add is not None else None, _ptr(add_scale), _ptr(work), B, C, T, Ks, Ls, Ss, float(eps), int(mode),
            _stream()), "ign_shapelet_bwd_bank")
        return gws
    for g, w in enumerate(ws):
        K, _, Lg = w.shape
        tstar, zmu, dsave, col0, stride, xstat = saved[g]
        nbytes = L.ign_shapelet_bwd_workspace_bytes(B, C, T, K, Lg, stride, mode)
        if nbytes == 0:
            raise _lib.IgnError(f"shapelet backward: no launch plan for K={K} L={Lg} stride={stride}")
        work = torch.empty(nbytes // 4, device=xn.device, dtype=torch.float32)
        _lib.check(L.ign_shapelet_bwd(_ptr(xn), _ptr(w), _ptr(gP), _ptr(P), _ptr(D), ld, col0,
                                      _ptr(tstar), _ptr(zmu), _ptr(dsave), _ptr(xstat), _ptr(wnorms[g]), _ptr(gws[g]), _ptr(work),
                                      B, C, T, K, Lg, stride, eps, mode, _stream()), "ign_shapelet_bwd")
        if gw_add is not None and gw_add[g] is not None:
            gws[g] = gws[g] + (gw_add[g] if add_scale is None else gw_add[g] * add_scale)
    return gws


def _threshold_grads(gP, P, ws, saved, C):
    """LTS: dP/dthr = sigma'(thr - m) = P(1-P), summed over the batch  (IGN/model/Shapelet.py:109)"""
    gt = (gP * P * (1 - P)).sum(0)
    return [gt[sv[3]:sv[3] + w.shape[0] * C].view(1, w.shape[0], C) for w, sv in zip(ws, saved)]


class ShapeletBankFn(torch.autograd.Function):
    """All length groups of a shapelet bank in one autograd node.

    forward(xn, eps, mode, stride_list, n_groups, w_0..w_{G-1}[, thr_0..thr_{G-1}]) -> (P, Dmin, Tstar), all
    (B, sum_g K_g*C) with the reference's feature order g*K*C + k*C + c (IGN/model/Shapelet.py:84,195-196).
    Dmin is non-differentiable (the training loss never reads it: IGN/exp/experiment_classification.py:325-329).
    Tstar (int32) is the WINDOW INDEX of the best match -- arg-max_t p for the RBF gate, arg-min_t d for LTS, first index on
    ties; the match covers samples [Tstar*stride, Tstar*stride + L) -- which is what the reference's shapelet plots need
    (IGN/utils/shapelet_util.py:153 recomputes it on the host by sliding every shapelet over every series).
    """

    @staticmethod
    def forward(ctx, xn, eps, mode, strides, n_groups, *params):
        ws = [w.contiguous() for w in params[:n_groups]]
        thrs = [t.contiguous() for t in params[n_groups:]] if (mode & GATE_LTS) else [None] * n_groups
        _need_gpu("shapelet_fwd", xn, *ws, *[t for t in thrs if t is not None])
        xn = xn.contiguous()
        need_grad = any(ctx.needs_input_grad[5:])     # grad mode is off inside forward(); ask the node instead
        P, D, saved = _bank_forward(xn, ws, thrs, eps, mode, strides, need_grad)
        Tstar = _cat_tstar(saved, xn.shape[0])
        ctx.mark_non_differentiable(D, Tstar)
        ctx.set_materialize_grads(False)              # (else autograd fills a zero tensor per unused output, one launch each)
        ctx.meta = (float(eps), int(mode), n_groups, saved, need_grad)
        ctx.save_for_backward(xn, P, D, *ws, *[t for t in thrs if t is not None])
        return P, D, Tstar

    @staticmethod
    def backward(ctx, gP, gD, gT):
        eps, mode, G, saved, had_grad = ctx.meta
        if not had_grad:
            raise _lib.IgnError("shapelet backward called but the forward ran without saving distances")
        xn, P, D = ctx.saved_tensors[:3]
        ws = ctx.saved_tensors[3:3 + G]
        if gP is None:
            return (None,) * (5 + G + (G if mode & GATE_LTS else 0))
        gP = gP.contiguous()
        grads_w = _bank_backward(xn, ws, gP, P, D, saved, eps, mode)
        grads_t = _threshold_grads(gP, P, ws, saved, xn.shape[1]) if mode & GATE_LTS else []
        return (None, None, None, None, None, *grads_w, *grads_t)


def shapelet_bank(xn, weights, eps, mode=DIST_L1 | GATE_RBF, strides=None, thresholds=None, return_tstar=False):
    """-> (P, Dmin) or, with return_tstar, (P, Dmin, Tstar): see ShapeletBankFn."""
    G = len(weights)
    if (mode & 0xf) == DIST_PEARSON:
        # pearson_corrcoef centres both operands (Shapelet.py:11-19).  <x - mean x, w_c> == <x, w_c> for a centred w_c,
        # so the kernel gets the centred shapelets and autograd projects the gradient back through this subtraction.
        weights = [w - w.mean(dim=-1, keepdim=True) for w in weights]
    strides = strides or [1] * G
    params = list(weights) + (list(thresholds) if (mode & GATE_LTS) else [])
    P, D, Tstar = ShapeletBankFn.apply(xn, eps, mode, tuple(strides), G, *params)
    return (P, D, Tstar) if return_tstar else (P, D)


class SbmFn(torch.autograd.Function):
    """The shapelet bottleneck model behind the instance norm as ONE autograd node: shapelet bank (every length group) ->
    linear class head (optional) -> both regularisers (IGN/model/Shapelet.py:190-210, 217-230).

    forward(xn, cfg, W, w_0..w_{G-1}[, thr_0..thr_{G-1}]) -> (P, Dmin, Tstar | None, reg, out | None)
      cfg = (eps, mode, strides, G, lambda_reg, lambda_div, fuse_head, want_tstar, reg_workspace)
      reg (1,) = lambda_reg * mean|W| + lambda_div * sum_g diversity_g -- value AND gradients from one launch
      (ign_sbm_reg_fwd_bwd); out = P W^T when fuse_head.
    Why one node: W and every w_g receive a gradient from the data path and one from a regulariser.  As separate nodes
    autograd adds the two with one accumulate kernel per parameter and the regularisers cost ~30 small launches per step
    (abs / mean / mul, four diversity launches, their sums and products, backward mirrors).  Here the head's weight-gradient
    kernel and the bank's reduction launch add `upstream * regulariser gradient` in their epilogues: the backward is
    head_bwd (2 launches) + G shapelet launches + 1 reduction, with no torch kernel in between."""

    @staticmethod
    def forward(ctx, xn, cfg, W, *params):
        eps, mode, strides, G, lam_reg, lam_div, fuse_head, want_tstar, reg_ws = cfg
        ws = [w.contiguous() for w in params[:G]]
        thrs = [t.contiguous() for t in params[G:]] if (mode & GATE_LTS) else [None] * G
        _need_gpu("sbm", xn, W, *ws, *[t for t in thrs if t is not None])
        xn, W = xn.contiguous(), W.contiguous()
        B, C, T = xn.shape
        need_grad = any(ctx.needs_input_grad[2:])
        L = _lib.lib()
        P, D, saved = _bank_forward(xn, ws, thrs, eps, mode, strides, need_grad)
        Tstar = _cat_tstar(saved, B) if want_tstar else None
        # regularisers: value + gradients, one launch
        use_div = lam_div > 0.0
        gWreg = torch.empty_like(W)
        gdiv = [torch.empty_like(w) for w in ws] if use_div else None
        reg = torch.empty(1, device=xn.device, dtype=torch.float32)
        Gd = G if use_div else 0
        _, ia = _tables(max(G, 1))
        _lib.check(L.ign_sbm_reg_fwd_bwd(_ptr(W), _ptr(gWreg), W.numel(), float(lam_reg), Gd, _pv(ws), _pv(gdiv) if use_div else None,
                                         ia(*[w.shape[0] for w in ws]), ia(*[w.shape[2] for w in ws]), C, float(lam_div), 1e-6,
                                         _ptr(reg), _ptr(reg_ws), _stream()), "ign_sbm_reg_fwd_bwd")
        out = None
        if fuse_head:
            N, F_ = W.shape
            out = torch.empty(B, N, device=xn.device, dtype=torch.float32)
            _lib.check(L.ign_head_fwd(_ptr(P), _ptr(W), None, _ptr(out), B, F_, N, P.stride(0), _stream()), "ign_head_fwd")
        ctx.set_materialize_grads(False)
        nd = [D] + ([Tstar] if Tstar is not None else [])
        ctx.mark_non_differentiable(*nd)
        ctx.meta = (float(eps), int(mode), G, saved, need_grad, fuse_head, gdiv)
        ctx.save_for_backward(xn, P, D, W, gWreg, *ws)
        return P, D, Tstar, reg, out

    @staticmethod
    def backward(ctx, gP, gD, gT, greg, gout):
        eps, mode, G, saved, had_grad, fuse_head, gdiv = ctx.meta
        if not had_grad:
            raise _lib.IgnError("shapelet backward called but the forward ran without saving distances")
        xn, P, D, W, gWreg = ctx.saved_tensors[:5]
        ws = ctx.saved_tensors[5:5 + G]
        B, C, T = xn.shape
        L = _lib.lib()
        greg = greg.contiguous().reshape(1) if greg is not None else None
        gW = None
        if fuse_head and gout is not None:
            N, F_ = W.shape
            gout = gout.contiguous()
            gPh = torch.empty_like(P)
            gW = torch.empty_like(W)
            _lib.check(L.ign_head_bwd_acc(_ptr(gout), _ptr(P), _ptr(W), _ptr(gPh), _ptr(gW), None,
                                          _ptr(gWreg) if greg is not None else None, _ptr(greg), B, F_, N, P.stride(0), _stream()),
                       "ign_head_bwd_acc")
            gP = gPh if gP is None else gPh + gP
        elif greg is not None and ctx.needs_input_grad[2]:
            gW = gWreg * greg
        if gP is None:             # nothing reached the gate outputs: only the regulariser moves the shapelets
            grads_w = [(gd * greg if (gdiv is not None and greg is not None) else None) for gd in (gdiv or [None] * G)]
            grads_t = [None] * G if mode & GATE_LTS else []
        else:
            gP = gP.contiguous()
            add = gdiv if (gdiv is not None and greg is not None) else None
            grads_w = _bank_backward(xn, ws, gP, P, D, saved, eps, mode, gw_add=add, add_scale=greg if add is not None else None)
            grads_t = _threshold_grads(gP, P, ws, saved, C) if mode & GATE_LTS else []
        return (None, None, gW, *grads_w, *grads_t)


def _bl_strides(t, name):
    """(B,L,H,E) tensor -> element strides of the batch and sequence axes; inner (H,E) block must be dense."""
    B, L, H, E = t.shape
    if t.stride(3) != 1 or t.stride(2) != E:
        raise _lib.IgnError(f"attention: {name} needs unit stride over E and stride E over H, got {t.stride()}")
    return t.stride(0), t.stride(1)


class AttentionFn(torch.autograd.Function):
    """softmax(scale * Q K^T) V with q (B,L,H,E), k/v (B,S,H,E) -> (B,L,H,E); IGN/layers/SelfAttention_Family.py:56-75."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        _need_gpu("attention", q, k, v)
        B, L, H, E = q.shape
        S = k.shape[1]
        fix = lambda t: t if (t.stride(3) == 1 and t.stride(2) == E and t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0
                              and t.data_ptr() % 16 == 0) else t.contiguous()
        q, k, v = fix(q), fix(k), fix(v)
        out = torch.empty(B, L, H, E, device=q.device, dtype=torch.float32)
        lse = torch.empty(B, H, L, device=q.device, dtype=torch.float32)
        (qb, ql), (kb, kl), (vb, vl) = _bl_strides(q, "q"), _bl_strides(k, "k"), _bl_strides(v, "v")
        L_ = _lib.lib()
        # "bf16x6": split-bf16 products on the bf16 matrix cores (fp32 accuracy); "f32": the fp32-MFMA kernel; inside an
        # autocast region (the reference's default mode): operands rounded to bf16, one product
        ctx.bf16 = torch.is_autocast_enabled() and E <= 64
        # "f16x3": two fp16 planes of power-of-two-scaled operands, three products (ign_attn_fwd_h3); needs the magnitude bounds
        ctx.h3 = (not ctx.bf16) and ATTN_MATH == "bf16x6" and GEMM_MATH == "f16x3" and E <= 64
        ctx.bounds = None
        if ctx.h3:
            ctx.bounds = (tensor_bound(q), tensor_bound(k), tensor_bound(v))
            _lib.check(L_.ign_attn_fwd_h3(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, L, S, H, E, qb, ql, kb, kl, vb, vl,
                                          float(scale), _stream(), *[_ptr(t) for t in ctx.bounds]), "ign_attn_fwd_h3")
        else:
            fwd = L_.ign_attn_fwd_bf16 if ctx.bf16 else (L_.ign_attn_fwd_x6 if ATTN_MATH == "bf16x6" else L_.ign_attn_fwd)
            _lib.check(fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, L, S, H, E, qb, ql, kb, kl, vb, vl, float(scale),
                           _stream()), "ign_attn_fwd")
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, gout):
        q, k, v, out, lse = ctx.saved_tensors
        B, L, H, E = q.shape
        S = k.shape[1]
        gout = gout.contiguous()
        gq = torch.empty(B, L, H, E, device=q.device, dtype=torch.float32)
        gk = torch.empty(B, S, H, E, device=q.device, dtype=torch.float32)
        gv = torch.empty_like(gk)
        delta = torch.empty(B, H, L, device=q.device, dtype=torch.float32)
        (qb, ql), (kb, kl), (vb, vl) = _bl_strides(q, "q"), _bl_strides(k, "k"), _bl_strides(v, "v")
        if ctx.h3:
            gmax = _new_slot(q.device)            # max over |dq|, |dk|, |dv| (taken as they are stored): one bound for the three
            _lib.check(_lib.lib().ign_attn_bwd_h3(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), _ptr(gout), _ptr(gq), _ptr(gk), _ptr(gv),
                                                  _ptr(delta), B, L, S, H, E, qb, ql, kb, kl, vb, vl, ctx.scale, _stream(), 0, 0,
                                                  *[_ptr(t) for t in ctx.bounds], _ptr(tensor_bound(gout)), _ptr(gmax)),
                       "ign_attn_bwd_h3")
            return set_bound(gq, gmax), set_bound(gk, gmax), set_bound(gv, gmax), None
        # split-bf16 kernels up to E = 64 (E = 128 exceeds their register budget: the fp32-MFMA backward is faster there)
        bwd = (_lib.lib().ign_attn_bwd_bf16 if ctx.bf16 else
               _lib.lib().ign_attn_bwd_x6 if (ATTN_MATH == "bf16x6" and E <= 64) else _lib.lib().ign_attn_bwd)
        _lib.check(bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), _ptr(gout), _ptr(gq), _ptr(gk), _ptr(gv), _ptr(delta),
                       B, L, S, H, E, qb, ql, kb, kl, vb, vl, ctx.scale, _stream()), "ign_attn_bwd")
        return gq, gk, gv, None


def attention(q, k, v, scale):
    out = AttentionFn.apply(q, k, v, scale)
    if GEMM_MATH == "f16x3" and ATTN_MATH == "bf16x6" and not torch.is_autocast_enabled() and q.shape[-1] <= 64:
        set_bound(out, tensor_bound(v))        # a row of the output is a convex combination of rows of v (cached: no extra pass)
    return out


class PackedAttentionFn(torch.autograd.Function):
    """Self-attention on a packed projection qkv (B, L, 3, H, E) -> (B, L, H, E): q / k / v are read as strided views and the
    three gradients are written straight into one packed buffer (ign_attn_bwd_x6_strided), so a fused q/k/v Linear layer gets
    its output gradient without a stack / gather pass."""

    @staticmethod
    def forward(ctx, qkv, scale):
        _need_gpu("attention", qkv)
        qkv = qkv.contiguous()
        B, L, three, H, E = qkv.shape
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        out = torch.empty(B, L, H, E, device=qkv.device, dtype=torch.float32)
        lse = torch.empty(B, H, L, device=qkv.device, dtype=torch.float32)
        sb, sl = qkv.stride(0), qkv.stride(1)
        L_ = _lib.lib()
        ctx.bf16 = torch.is_autocast_enabled()
        ctx.h3 = (not ctx.bf16) and GEMM_MATH == "f16x3"
        ctx.bound = None
        if ctx.h3:
            ctx.bound = tensor_bound(qkv)             # one pass: a bound of the packed tensor bounds q, k and v alike
            bp = _ptr(ctx.bound)
            _lib.check(L_.ign_attn_fwd_h3(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, L, L, H, E, sb, sl, sb, sl, sb, sl,
                                          float(scale), _stream(), bp, bp, bp), "ign_attn_fwd_h3")
        else:
            fwd = L_.ign_attn_fwd_bf16 if ctx.bf16 else L_.ign_attn_fwd_x6
            _lib.check(fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, L, L, H, E, sb, sl, sb, sl, sb, sl, float(scale),
                           _stream()), "ign_attn_fwd_x6")
        ctx.save_for_backward(qkv, out, lse)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, gout):
        qkv, out, lse = ctx.saved_tensors
        B, L, three, H, E = qkv.shape
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
        gout = gout.contiguous()
        gqkv = torch.empty_like(qkv)
        delta = torch.empty(B, H, L, device=qkv.device, dtype=torch.float32)
        sb, sl = qkv.stride(0), qkv.stride(1)
        if ctx.h3:
            bp = _ptr(ctx.bound)
            gmax = _new_slot(qkv.device)          # max |dq|, |dk|, |dv|, taken by the kernels as they store: the projection's
            _lib.check(_lib.lib().ign_attn_bwd_h3(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), _ptr(gout), _ptr(gqkv[:, :, 0]),
                                                  _ptr(gqkv[:, :, 1]), _ptr(gqkv[:, :, 2]), _ptr(delta), B, L, L, H, E, sb, sl, sb, sl,
                                                  sb, sl, ctx.scale, _stream(), sb, sl, bp, bp, bp, _ptr(tensor_bound(gout)),
                                                  _ptr(gmax)), "ign_attn_bwd_h3")
            return set_bound(gqkv, gmax), None    # backward GEMMs scale their dL/dy operand by it without a pass over it
        _lib.check(_lib.lib().ign_attn_bwd_x6_strided(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), _ptr(gout),
                                                      _ptr(gqkv[:, :, 0]), _ptr(gqkv[:, :, 1]), _ptr(gqkv[:, :, 2]), _ptr(delta),
                                                      B, L, L, H, E, sb, sl, sb, sl, sb, sl, ctx.scale, _stream(), sb, sl,
                                                      1 if ctx.bf16 else 0), "ign_attn_bwd_x6_strided")
        return gqkv, None


def attention_packed(qkv, scale):
    """softmax(scale q k^T) v for qkv (B, L, 3, H, E); the split-bf16 kernels cover E <= 64 (else the unpacked path)."""
    E = qkv.shape[-1]
    if ATTN_MATH != "bf16x6" or E > 64:
        return attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], scale)
    out = PackedAttentionFn.apply(qkv, scale)
    if GEMM_MATH == "f16x3" and not torch.is_autocast_enabled():
        set_bound(out, tensor_bound(qkv))      # convex combinations of rows of v: bounded by the packed tensor's bound (cached)
    return out


class HeadLinearFn(torch.autograd.Function):
    """Skinny expert-head GEMM x (B,F) @ W(N,F)^T + bias -> (B,N) on ign_head_fwd/bwd."""

    @staticmethod
    def forward(ctx, x, w, bias):
        _need_gpu("head_linear", x, w, bias)
        x = x if (x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0) else x.contiguous()
        w = w.contiguous()
        B, F_ = x.shape
        N = w.shape[0]
        out = torch.empty(B, N, device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().ign_head_fwd(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), B, F_, N, x.stride(0), _stream()),
                   "ign_head_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        B, F_ = x.shape
        N = w.shape[0]
        g = g.contiguous()
        gx = torch.empty(B, F_, device=x.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        gb = torch.empty(N, device=x.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if gb is not None and gw is None:
            gw = torch.empty_like(w)
        _lib.check(_lib.lib().ign_head_bwd(_ptr(g), _ptr(x), _ptr(w), _ptr(gx), _ptr(gw), _ptr(gb), B, F_, N, x.stride(0),
                                           _stream()), "ign_head_bwd")
        # gx was written with row pitch x.stride(0); it is dense only if x was (the common case)
        if gx is not None and x.stride(0) != F_:
            raise _lib.IgnError("head_linear backward: padded input rows are not supported")
        return gx, (gw if ctx.needs_input_grad[1] else None), gb


def head_linear(x, w, bias=None):
    """nn.Linear with few outputs (class logits).  Shapes the streaming kernel does not cover go to torch's GEMM
    (still on the GPU; CPU tensors are refused)."""
    if not x.is_cuda:
        raise _lib.IgnError(f"head_linear: tensor on {x.device}; the product path runs on the MI355X only (no CPU fallback)")
    if x.dim() != 2 or x.shape[1] % 4 or w.shape[0] > 16 or x.dtype != torch.float32 \
            or w.dtype != torch.float32 or torch.is_autocast_enabled():
        return torch.nn.functional.linear(x, w, bias)
    return HeadLinearFn.apply(x, w, bias)


# Arithmetic of the dense-layer / convolution GEMMs outside an autocast region (include/ign_abi.h, "h3"):
#   "f16x3"  (default) operands scaled by a power of two from a device-side magnitude bound, split into two fp16 planes, three
#            products -- fp32-level accuracy against float64 like "bf16x6", at 1.5-1.6x its speed;
#   "bf16x6" three bf16 planes, six products.
GEMM_MATH = os.environ.get("IGN_GEMM_MATH", "f16x3")

_SLOT_POOL = {}


def _new_slot(device):
    """One zero-initialised float on `device`, cut from a pooled zero buffer (one fill launch per 256 slots).  While a hipGraph is
    being captured the slot gets its own fill INSIDE the graph: the kernels max into it atomically, so a slot zeroed once at
    allocation would carry the maximum over all earlier replays into every later one."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(1, device=device, dtype=torch.float32)
    pool = _SLOT_POOL.get(device)
    if pool is None or pool[1] >= pool[0].numel():
        pool = _SLOT_POOL[device] = [torch.zeros(256, device=device, dtype=torch.float32), 0]
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1]


def cached_bound(t):
    """The bound attached to `t` (one-element device tensor) if it still describes the tensor's contents, else None.  A bound
    that was MEASURED from the data in an earlier call (ign_absmax, ign_instnorm_fwd_amax) is not trusted while a hipGraph is
    being captured: the capture would bake in 'already known' and replay the example batch's bound for every later batch
    (GraphedTrainStep refills the same input tensors).  Bounds attached by a producer inside the captured region carry the
    flag `in_graph` and stay valid -- their kernels are part of the graph."""
    cb = getattr(t, "_ign_bound", None)
    if cb is None or cb[1] != t._version or cb[2] != t.data_ptr():
        return None
    if torch.cuda.is_current_stream_capturing() and not (len(cb) > 3 and cb[3]):
        return None
    return cb[0]


def tensor_bound(t):
    """A one-element device tensor holding an upper bound of max |t| -- what the fp16 GEMMs scale their operands by.  A producer
    that knows a bound attaches it (`set_bound`: LayerNorm's hard bound from its parameters); otherwise ONE pass over the tensor
    takes the exact maximum (ign_absmax), cached on the tensor object for as long as it is not modified in place."""
    # Parameters are never cached: the flat Adam kernel (ign_adam_step) rewrites them through raw pointers, which does not move
    # the version counter a cache entry is validated by -- a stale bound would let a weight that has grown overflow fp16.
    param = t.is_leaf and t.requires_grad
    cached = None if param else cached_bound(t)
    if cached is not None:
        return cached
    base = None if param else getattr(t, "_base", None)
    if base is not None and (base.is_leaf and base.requires_grad):
        base = None
    if base is not None and base.numel() == t.numel():       # a reshaped view of a tensor whose producer attached a bound
        cb = cached_bound(base)
        if cb is not None:
            return cb
    if not t.is_contiguous():
        base = getattr(t, "_base", None)
        if base is not None and base.is_contiguous() and base.dtype == t.dtype and base.numel() <= 4 * t.numel():
            return tensor_bound(base)         # a strided view (q / k / v of a packed projection): the base tensor's bound bounds it
    tc = t if t.is_contiguous() else t.contiguous()
    slot = _new_slot(t.device)
    _lib.check(_lib.lib().ign_absmax(_ptr(tc), tc.numel(), _ptr(slot), _stream()), "ign_absmax")
    if not param:
        set_bound(t, slot)
    return slot


def set_bound(t, slot):
    """Attach a known magnitude bound (one-element device tensor) to `t`; see tensor_bound / cached_bound.  A reshaped view is
    looked up through its ROOT base (`view(...).reshape(...)._base` is the root, not the intermediate view), so a bound attached
    to a full-size view is attached to that root as well."""
    cap = torch.cuda.is_current_stream_capturing()
    try:
        t._ign_bound = (slot, t._version, t.data_ptr(), cap)
        base = t._base
        if base is not None and base.numel() == t.numel() and not (base.is_leaf and base.requires_grad):
            base._ign_bound = (slot, base._version, base.data_ptr(), cap)
    except Exception:
        pass
    return t


def keep_bound(out, src, factor=1.0):
    """`out` = f(src) element-wise with |f(u)| <= factor |u|: `out` inherits src's magnitude bound, if src carries one, instead of
    being scanned when a dense layer consumes it.  factor 1: ReLU, GELU.  factor <= 2 (GELU's derivative reaches 1.13) is
    covered by the headroom of the scaling: a bound b is scaled to [2^13, 2^14), fp16 represents values below 2^16, so an
    element up to 4 b neither overflows nor loses a bit (include/ign_abi.h, "h3")."""
    cb = cached_bound(src) if factor <= 2.0 else None
    if cb is not None:
        set_bound(out, cb)
    return out


class _ActFn(torch.autograd.Function):
    """ReLU / GELU with torch's own element-wise kernels in both directions; what it adds is the magnitude bound: the output
    inherits the input's, and dL/du inherits dL/dy's (|relu'| <= 1, |gelu'| <= 1.13) -- a plain F.relu / F.gelu node hands the
    backward GEMM behind it a tensor without one, i.e. costs a pass over it (ign_absmax)."""

    @staticmethod
    def forward(ctx, u, gelu):
        ctx.gelu = gelu
        y = torch.nn.functional.gelu(u) if gelu else torch.relu(u)
        ctx.save_for_backward(u if gelu else y)
        return y

    @staticmethod
    def backward(ctx, g):
        (t,) = ctx.saved_tensors
        gu = torch.ops.aten.gelu_backward(g, t) if ctx.gelu else torch.ops.aten.threshold_backward(g, t, 0)
        return keep_bound(gu, g, 1.13 if ctx.gelu else 1.0), None


def relu(u):
    return keep_bound(_ActFn.apply(u, False), u)


def gelu(u):
    return keep_bound(_ActFn.apply(u, True), u)


def _gemm(L, bf16):
    """(forward / input-gradient GEMM, weight-gradient GEMM) entry points: the split-bf16 kernels (fp32 accuracy), or inside a
    torch.autocast(bfloat16) region -- the reference's default mode -- the same kernels with their operands rounded to bf16 and
    ONE product per MFMA step (what autocast computes for a matmul; accumulation and outputs stay fp32)."""
    return (L.ign_clconv_fwd_bf16, L.ign_clconv_wgrad_bf16) if bf16 else (L.ign_clconv_fwd_x6, L.ign_clconv_wgrad_x6)


_PREPARED = {}          # (data_ptr, shape) -> (w, version, generation, bound slot, packed forward planes, packed transposed planes)


def prepare_linear_weights(weights, need_dx=True):
    """Model-level prologue of the fp16-plane dense layers: the magnitude bounds of ALL the given weight matrices in one launch
    (ign_absmax_multi) and their packed plane forms in one launch per eight (ign_clconv_pack_weights_h2_multi), instead of one
    scan + one packing launch per layer.  `ops.linear` consumes the entry of its weight (once; an entry is valid only for the
    parameter values it was made from: tensor version and _lib.PARAM_GENERATION).  No-op outside the f16x3 arithmetic."""
    _PREPARED.clear()
    if GEMM_MATH != "f16x3" or torch.is_autocast_enabled():
        return
    ws = []
    for w in weights:
        if w is None or not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or w.shape[0] % 4 or not w.is_contiguous():
            continue
        ws.append(w)
    if not ws:
        return
    L = _lib.lib()
    dev = ws[0].device
    slots = [_new_slot(dev) for _ in ws]
    for c0 in range(0, len(ws), 16):
        cw, cs = ws[c0:c0 + 16], slots[c0:c0 + 16]
        n = len(cw)
        vpa, lla = ctypes.c_void_p * n, ctypes.c_longlong * n
        _lib.check(L.ign_absmax_multi(n, vpa(*[w.data_ptr() for w in cw]), lla(*[w.numel() for w in cw]),
                                      vpa(*[t.data_ptr() for t in cs]), _stream()), "ign_absmax_multi")
    gen = _lib.PARAM_GENERATION[0]
    for c0 in range(0, len(ws), 8):
        cw, cs = ws[c0:c0 + 8], slots[c0:c0 + 8]
        n = len(cw)
        vpa, ia = ctypes.c_void_p * n, ctypes.c_int * n
        wt = [torch.empty(int(L.ign_clconv_x3_elems(w.shape[0], w.shape[1], 1)), device=dev, dtype=torch.bfloat16) for w in cw]
        wd = [torch.empty(int(L.ign_clconv_x3_elems(w.shape[1], w.shape[0], 1)), device=dev, dtype=torch.bfloat16) if need_dx else None
              for w in cw]
        _lib.check(L.ign_clconv_pack_weights_h2_multi(n, vpa(*[w.data_ptr() for w in cw]), vpa(*[t.data_ptr() for t in wt]),
                                                      vpa(*[(t.data_ptr() if t is not None else None) for t in wd]) if need_dx else None,
                                                      ia(*[w.shape[0] for w in cw]), ia(*[w.shape[1] for w in cw]), ia(*([1] * n)), None,
                                                      vpa(*[t.data_ptr() for t in cs]), _stream()), "ign_clconv_pack_weights_h2_multi")
        for w, sl, a, b in zip(cw, cs, wt, wd):
            _PREPARED[(w.data_ptr(), tuple(w.shape))] = (w, w._version, gen, sl, a, b)


def _take_prepared(w, need_dx):
    ent = _PREPARED.pop((w.data_ptr(), tuple(w.shape)), None)
    if ent is None or ent[1] != w._version or ent[2] != _lib.PARAM_GENERATION[0] or (need_dx and ent[5] is None):
        return None
    return ent


def _linear_forward(ctx, x, w, bias, extra=()):
    """LinearFn.forward; `extra`: tensors saved behind x2 (GeluLinearFn: the pre-activation)."""
    L = _lib.lib()
    Co, Ci = w.shape
    x2 = x.reshape(-1, Ci)
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    M = x2.shape[0]
    w = w.contiguous()
    dev = x.device
    need_dx = ctx.needs_input_grad[0]
    ctx.bf16 = torch.is_autocast_enabled()          # autocast region: operands rounded to bf16, one product (see _gemm)
    ctx.h3 = (not ctx.bf16) and GEMM_MATH == "f16x3"
    prepared = _take_prepared(w, need_dx) if ctx.h3 else None       # bound + packed planes from the model's prologue launch
    if prepared is not None:
        wt3, wd3 = prepared[4], (prepared[5] if need_dx else None)
    else:
        wt3 = torch.empty(int(L.ign_clconv_x3_elems(Co, Ci, 1)), device=dev, dtype=torch.bfloat16)
        wd3 = torch.empty(int(L.ign_clconv_x3_elems(Ci, Co, 1)), device=dev, dtype=torch.bfloat16) if need_dx else None
    y = torch.empty(*x.shape[:-1], Co, device=dev, dtype=torch.float32)    # final shape (not a view: its reshaped views find its bound)
    ctx.bx = ctx.bw = None
    if ctx.h3:
        # two fp16 planes, three products: operand bounds on the device (the input's is inherited from x when x2 is a view)
        ctx.bx = tensor_bound(x) if x2.data_ptr() == x.data_ptr() and x.is_contiguous() else tensor_bound(x2)
        if prepared is not None:
            ctx.bw = prepared[3]
        else:
            ctx.bw = tensor_bound(w)
            v1, i1 = _tables(1)
            _lib.check(L.ign_clconv_pack_weights_h2_multi(1, v1(w.data_ptr()), v1(wt3.data_ptr()),
                                                          v1(wd3.data_ptr()) if need_dx else None, i1(Co), i1(Ci), i1(1), None,
                                                          v1(ctx.bw.data_ptr()), _stream()), "ign_clconv_pack_weights_h2_multi")
        # the epilogue also takes max |y|: the operand bound of whatever dense layer / attention core consumes y
        yb = _new_slot(dev)
        _lib.check(L.ign_clconv_fwd_h3_amax(_ptr(x2), _ptr(wt3), _ptr(bias), None, None, _ptr(y), None, _ptr(ctx.bx), _ptr(ctx.bw),
                                            _ptr(yb), 1, M, Ci, Co, 1, _stream()), "ign_clconv_fwd_h3")
    else:
        yb = None
        _lib.check(L.ign_clconv_pack_weights_x3(_ptr(w), _ptr(wt3), _ptr(wd3), Co, Ci, 1, _stream()), "ign_clconv_pack_weights_x3")
        _lib.check(_gemm(L, ctx.bf16)[0](_ptr(x2), _ptr(wt3), _ptr(bias), None, None, _ptr(y), None, 1, M, Ci, Co, 1, _stream()),
                   "ign_clconv_fwd_x6")
    ctx.save_for_backward(x2, *extra)
    ctx.wd3, ctx.dims, ctx.has_bias, ctx.xshape = wd3, (M, Ci, Co), bias is not None, x.shape
    ctx.mark_non_differentiable(*([yb] if yb is not None else []))
    ctx.set_materialize_grads(False)              # no zero-filled "gradient" of the bound output per backward call
    return y, yb


def _linear_backward(ctx, gy, gelu=False):
    """LinearFn.backward.  `gelu`: the layer's input was gelu(u), u saved behind x2 -- the returned input gradient is dL/du, the
    activation's derivative applied in the epilogue of the input-gradient GEMM (ign_linear_dgrad_gelu_h3) where the shape allows."""
    if gy is None:
        return None, None, None
    L = _lib.lib()
    x2 = ctx.saved_tensors[0]
    u2 = ctx.saved_tensors[1].reshape(x2.shape) if gelu else None
    M, Ci, Co = ctx.dims
    g2 = gy.reshape(M, Co)
    g2 = g2 if g2.is_contiguous() else g2.contiguous()
    dx = dw = db = None
    bg = tensor_bound(g2) if ctx.h3 else None
    if ctx.needs_input_grad[0]:
        dx = torch.empty(M, Ci, device=g2.device, dtype=torch.float32)
        dxb = None
        fused = gelu and ctx.h3 and Ci % 256 == 0 and Co % 4 == 0 and u2.is_contiguous()
        if fused:
            dxb = _new_slot(g2.device)           # dL/du = (g W) * gelu'(u) in one GEMM; its epilogue also takes max |dL/du|
            _lib.check(L.ign_linear_dgrad_gelu_h3(_ptr(g2), _ptr(ctx.wd3), _ptr(u2), _ptr(dx), _ptr(bg), _ptr(ctx.bw), _ptr(dxb), M, Co, Ci,
                                                  _stream()), "ign_linear_dgrad_gelu_h3")
        elif ctx.h3:
            dxb = _new_slot(g2.device)           # the epilogue takes max |dx|: the next backward GEMM's operand bound
            _lib.check(L.ign_clconv_fwd_h3_amax(_ptr(g2), _ptr(ctx.wd3), None, None, None, _ptr(dx), None, _ptr(bg), _ptr(ctx.bw),
                                                _ptr(dxb), 1, M, Co, Ci, 1, _stream()), "ign_clconv_fwd_h3_amax(dx)")
        else:
            _lib.check(_gemm(L, ctx.bf16)[0](_ptr(g2), _ptr(ctx.wd3), None, None, None, _ptr(dx), None, 1, M, Co, Ci, 1, _stream()),
                       "ign_clconv_fwd_x6(dx)")
        if gelu and not fused:                   # shapes / arithmetics outside the fused kernel: torch's element-wise backward
            dx = torch.ops.aten.gelu_backward(dx, u2)     # (|gelu'| <= 1.13: dxb stays a usable bound, see keep_bound)
        dx = dx.view(ctx.xshape)
        if dxb is not None:
            set_bound(dx, dxb)
    want_db = ctx.has_bias and ctx.needs_input_grad[2]
    if ctx.needs_input_grad[1]:
        dw = torch.empty(Co, Ci, device=g2.device, dtype=torch.float32)
        if Ci % 4 == 0 and LINEAR_WGRAD == "bf16x6":
            # weight and bias gradient in one pass over dy (the bias gradient rides on the tiles that stage dy anyway)
            ws = torch.empty(int(L.ign_clconv_wgrad_x6_workspace_bytes(1, M, Ci, Co, 1)) // 4, device=g2.device,
                             dtype=torch.float32)
            if want_db:
                db = torch.empty(Co, device=g2.device, dtype=torch.float32)
            if ctx.h3:
                _lib.check(L.ign_linear_wgrad_h3(_ptr(g2), _ptr(x2), _ptr(dw), _ptr(db), _ptr(ws), _ptr(bg), _ptr(ctx.bx), M, Ci, Co,
                                                 _stream()), "ign_linear_wgrad_h3")
            else:
                fn = L.ign_linear_wgrad_bf16 if ctx.bf16 else L.ign_linear_wgrad_x6
                _lib.check(fn(_ptr(g2), _ptr(x2), _ptr(dw), _ptr(db), _ptr(ws), M, Ci, Co, _stream()), "ign_linear_wgrad_x6")
        else:
            ws = torch.empty(int(L.ign_clconv_wgrad_workspace_bytes(1, M, Ci, Co, 1)) // 4, device=g2.device,
                             dtype=torch.float32)
            _lib.check(L.ign_clconv_wgrad(_ptr(g2), 0, _ptr(x2), None, None, _ptr(dw), _ptr(ws), 1, M, Ci, Co, 1, _stream()),
                       "ign_clconv_wgrad")
    if want_db and db is None:
        db = g2.sum(dim=0)
    return dx, dw, db


class LinearFn(torch.autograd.Function):
    """y = x W^T + b for the dense layers of the two encoder baselines (IGN/layers/SelfAttention_Family.py:195-211,
    IGN/layers/Transformer_EncDec.py:33-48, nn.TransformerEncoderLayer in IGN/model/eegcnn.py:219-228) on the library's own
    GEMM kernels instead of hipBLASLt: a Linear layer is the k = 1 case of the channels-last convolution, so forward and the
    input gradient run on the split-bf16 kernel (fp32 accuracy on the bf16 matrix cores, ign_clconv_fwd_x6 with the weight
    resp. its transpose), and so does the weight gradient (ign_clconv_wgrad_x6, k = 1: 128 x 128 tiles, transposing LDS reads);
    `LINEAR_WGRAD = "f32"` (tests) keeps it on the fp32-MFMA TN kernel (ign_clconv_wgrad)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        return _linear_forward(ctx, x, w, bias)

    @staticmethod
    def backward(ctx, gy, _gyb=None):
        return _linear_backward(ctx, gy)


class GeluLinearFn(torch.autograd.Function):
    """z = gelu(u) W^T + b -- the activation and second dense layer of an encoder's feed-forward block
    (IGN/layers/Transformer_EncDec.py:46-47) as one node: forward = torch's GELU kernel + the dense layer's GEMM; backward = the
    weight gradient from the saved gelu(u) and the input gradient THROUGH the activation in one GEMM (ign_linear_dgrad_gelu_h3:
    dL/dy is never written, aten::gelu_backward's three passes over the (rows, d_ff) tensors disappear)."""

    @staticmethod
    def forward(ctx, u, w, bias):
        y = keep_bound(torch.nn.functional.gelu(u), u)
        return _linear_forward(ctx, y, w, bias, extra=(u,))

    @staticmethod
    def backward(ctx, gy, _gyb=None):
        return _linear_backward(ctx, gy, gelu=True)


def linear(x, w, bias=None):
    """nn.Linear / 1x1 Conv1d on the hand-written GEMM kernels (inside an autocast region: their single-product bf16 form);
    shapes they do not cover (and non-fp32 tensors) go to torch's GEMM."""
    if not x.is_cuda:
        raise _lib.IgnError(f"linear: tensor on {x.device}; the product path runs on the MI355X only (no CPU fallback)")
    if (x.dtype != torch.float32 or w.dtype != torch.float32
            or w.shape[0] % 4 or x.numel() == 0 or x.shape[-1] != w.shape[1]
            or x.numel() // x.shape[-1] >= (1 << 30)):
        return torch.nn.functional.linear(x, w, bias)
    y, yb = LinearFn.apply(x, w, bias)
    if yb is not None:
        set_bound(y, yb)
    return y


def gelu_linear(u, w, bias=None):
    """linear(gelu(u), w, bias) with the GELU's backward folded into the dense layer's input-gradient GEMM (GeluLinearFn); inputs the
    hand-written GEMMs do not cover take the two-op route."""
    if (not u.is_cuda or u.dtype != torch.float32 or w.dtype != torch.float32 or w.shape[0] % 4 or u.numel() == 0
            or u.shape[-1] != w.shape[1] or u.numel() // u.shape[-1] >= (1 << 30)):
        return linear(gelu(u), w, bias)
    y, yb = GeluLinearFn.apply(u, w, bias)
    if yb is not None:
        set_bound(y, yb)
    return y


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dimension on ign_layernorm_fwd / _bwd (HBM-bound row kernels; torch's kernels for narrow rows
    -- PatchTST's 3.9 M rows of 64 -- run 10x off the memory roofline)."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, eps):
        """`res` (same shape as x, or None): the row normalised is x + res -- the residual connection of a post-norm encoder
        layer -- added inside the same pass (ign_layernorm_res_fwd)."""
        _need_gpu("layer_norm", x, res, weight, bias)
        L = _lib.lib()
        D = x.shape[-1]
        x2 = x.reshape(-1, D)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        R = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(R, device=x.device, dtype=torch.float32)
        rstd = torch.empty(R, device=x.device, dtype=torch.float32)
        if res is not None:
            r2 = res.reshape(-1, D)
            r2 = r2 if r2.is_contiguous() else r2.contiguous()
            if r2.shape != x2.shape:
                raise _lib.IgnError(f"layer_norm: residual {tuple(res.shape)} does not match {tuple(x.shape)}")
            s2 = torch.empty_like(x2)
            _lib.check(L.ign_layernorm_res_fwd(_ptr(x2), _ptr(r2), _ptr(s2), _ptr(weight), _ptr(bias), _ptr(y), _ptr(mean), _ptr(rstd),
                                               R, D, float(eps), _stream()), "ign_layernorm_res_fwd")
            x2 = s2
        else:
            _lib.check(L.ign_layernorm_fwd(_ptr(x2), _ptr(weight), _ptr(bias), _ptr(y), _ptr(mean), _ptr(rstd), R, D, float(eps),
                                           _stream()), "ign_layernorm_fwd")
        ctx.save_for_backward(x2, weight, mean, rstd)
        ctx.has_bias = bias is not None
        ctx.has_res = res is not None
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        x2, weight, mean, rstd = ctx.saved_tensors
        R, D = x2.shape
        g2 = gy.reshape(R, D)
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        gx = torch.empty_like(x2)
        need_w = ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3])
        dgamma = torch.empty(D, device=x2.device, dtype=torch.float32) if need_w else None
        dbeta = torch.empty(D, device=x2.device, dtype=torch.float32) if need_w and ctx.has_bias else None
        part = torch.empty(int(L.ign_layernorm_parts(R, D)) * 2 * D, device=x2.device, dtype=torch.float32)
        gxv = gx.view(gy.shape)
        if GEMM_MATH == "f16x3" and not torch.is_autocast_enabled():
            # dL/dx usually feeds the backward GEMMs of a dense layer: its magnitude bound is taken here, as it is written
            slot = _new_slot(x2.device)
            _lib.check(L.ign_layernorm_bwd_amax(_ptr(x2), _ptr(g2), _ptr(weight), _ptr(mean), _ptr(rstd), _ptr(gx), _ptr(dgamma),
                                                _ptr(dbeta), _ptr(part), _ptr(slot), R, D, _stream()), "ign_layernorm_bwd_amax")
            set_bound(gxv, slot)
        else:
            _lib.check(L.ign_layernorm_bwd(_ptr(x2), _ptr(g2), _ptr(weight), _ptr(mean), _ptr(rstd), _ptr(gx), _ptr(dgamma), _ptr(dbeta),
                                           _ptr(part), R, D, _stream()), "ign_layernorm_bwd")
        return gxv, (gxv if ctx.has_res else None), dgamma, dbeta, None          # d(x + res): the same gradient for both addends


def layer_norm(x, norm, residual=None):
    """Apply an nn.LayerNorm module (normalised over the last dimension, affine) on the hand-written kernels; shapes they do not
    cover go through the module itself (torch on the GPU).  `residual`: normalise x + residual (the add rides on the same pass)."""
    D = x.shape[-1]
    # torch's kernels fall off the memory roofline for many narrow rows (Transformer: 256 000 rows of 512, 96.9 -> 94.9 ms/step;
    # PatchTST: 3.9 M rows of 64, 123.9 -> 79.4 ms/step); for few rows (EEG-CNN: 25 600 rows of 512) the grid is sized by the row
    # count and the d(gamma) partials are reduced in parallel: 22 / 37 us per call against torch's 34 / 97 (7.52 -> 7.36 ms/step)
    if (not x.is_cuda or x.dtype != torch.float32 or norm.weight is None or len(norm.normalized_shape) != 1 or D % 4 or D > 2048
            or x.numel() == 0 or x.numel() < LAYERNORM_MIN_ROWS * D
            or (residual is not None and (residual.shape != x.shape or residual.dtype != x.dtype or not residual.is_cuda))):
        return norm(x if residual is None else x + residual)
    out = LayerNormFn.apply(x, residual, norm.weight, norm.bias, norm.eps)
    if GEMM_MATH == "f16x3" and norm.bias is not None and not torch.is_autocast_enabled():
        # a row standardised with its own mean and (biased) variance over D elements cannot exceed sqrt(D - 1): the output is
        # bounded by max_d(|gamma_d| sqrt(D - 1) + |beta_d|) -- from the parameters alone, one tiny launch instead of a pass over
        # the activations when a dense layer behind it asks for its operand's magnitude (tensor_bound)
        slots = torch.empty(4, device=x.device, dtype=torch.float32)
        v1, i1, l1 = ctypes.c_void_p * 1, ctypes.c_int * 1, ctypes.c_longlong * 1
        wp, bp = norm.weight.data_ptr(), norm.bias.data_ptr()
        _lib.check(_lib.lib().ign_fcn_scan(1, v1(wp), l1(D), v1(wp), v1(bp), i1(D), l1(max(D, 2)), _ptr(slots), None, 0, _stream()), "ign_fcn_scan")
        set_bound(out, slots[1:2])
    return out


class ConvCLFn(torch.autograd.Function):
    """Valid, stride-1 Conv1d on a CHANNELS-LAST tensor: x (B, Tin, Ci), w (Co, Ci, k) -> y (B, Tin-k+1, Co).

    The general-purpose door to the implicit-GEMM kernels of the FCN expert (csrc/ign_clconv_x6.hip) for the other deep
    experts and embeddings (IGN/model/ResNet.py:11-22,46; IGN/layers/Embed.py:32-36): padding is materialised by the caller
    (zero rows, circular rows), a strided convolution is a stride-1 one over a space-to-depth view.  Forward and input
    gradient run on ign_clconv_fwd_x6 (the latter on dy zero-padded by k-1 rows with the tap-reversed transposed weights),
    the weight gradient on ign_clconv_wgrad_x6 where its tap count is instantiated (k in 1,2,3,5,8), else ign_clconv_wgrad."""

    @staticmethod
    def forward(ctx, x, w, bias):
        L = _lib.lib()
        _need_gpu("conv1d_cl", x, w)
        B, Tin, Ci = x.shape
        Co, Ci2, k = w.shape
        if Ci2 != Ci or Tin < k or Co % 4:
            raise _lib.IgnError(f"conv1d_cl: x {tuple(x.shape)} / w {tuple(w.shape)}: needs matching channels, Tin >= k, Co % 4 == 0")
        x = x.contiguous()
        w = w.contiguous()
        dev = x.device
        need_dx = ctx.needs_input_grad[0]
        wt3 = torch.empty(int(L.ign_clconv_x3_elems(Co, Ci, k)), device=dev, dtype=torch.bfloat16)
        wd3 = torch.empty(int(L.ign_clconv_x3_elems(Ci, Co, k)), device=dev, dtype=torch.bfloat16) if need_dx else None
        y = torch.empty(B, Tin - k + 1, Co, device=dev, dtype=torch.float32)
        ctx.bf16 = torch.is_autocast_enabled()
        ctx.h3 = (not ctx.bf16) and GEMM_MATH == "f16x3"
        ctx.bx = ctx.bw = None
        if ctx.h3:          # two fp16 planes, three products (see LinearFn)
            ctx.bw, ctx.bx = tensor_bound(w), tensor_bound(x)
            v1, i1 = _tables(1)
            _lib.check(L.ign_clconv_pack_weights_h2_multi(1, v1(w.data_ptr()), v1(wt3.data_ptr()), v1(wd3.data_ptr()) if need_dx else None,
                                                          i1(Co), i1(Ci), i1(k), None, v1(ctx.bw.data_ptr()), _stream()),
                       "ign_clconv_pack_weights_h2_multi")
            _lib.check(L.ign_clconv_fwd_h3(_ptr(x), _ptr(wt3), _ptr(bias), None, None, _ptr(y), None, _ptr(ctx.bx), _ptr(ctx.bw), B, Tin, Ci,
                                           Co, k, _stream()), "ign_clconv_fwd_h3")
        else:
            _lib.check(L.ign_clconv_pack_weights_x3(_ptr(w), _ptr(wt3), _ptr(wd3), Co, Ci, k, _stream()), "ign_clconv_pack_weights_x3")
            _lib.check(_gemm(L, ctx.bf16)[0](_ptr(x), _ptr(wt3), _ptr(bias), None, None, _ptr(y), None, B, Tin, Ci, Co, k, _stream()),
                       "ign_clconv_fwd_x6")
        ctx.save_for_backward(x)
        ctx.wd3, ctx.dims, ctx.has_bias = wd3, (B, Tin, Ci, Co, k), bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _lib.lib()
        (x,) = ctx.saved_tensors
        B, Tin, Ci, Co, k = ctx.dims
        Tout = Tin - k + 1
        dx = dw = db = None
        # dy zero-padded by k-1 rows per side: the operand of the input gradient, and the layout the weight gradient reads
        gyp = torch.nn.functional.pad(gy, (0, 0, k - 1, k - 1)) if k > 1 else gy.contiguous()
        bg = tensor_bound(gyp) if ctx.h3 else None            # (the zero rows do not change the maximum)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, Tin, Ci, device=gy.device, dtype=torch.float32)
            if ctx.h3:
                _lib.check(L.ign_clconv_fwd_h3(_ptr(gyp), _ptr(ctx.wd3), None, None, None, _ptr(dx), None, _ptr(bg), _ptr(ctx.bw), B,
                                               Tout + 2 * (k - 1), Co, Ci, k, _stream()), "ign_clconv_fwd_h3(dx)")
            else:
                _lib.check(_gemm(L, ctx.bf16)[0](_ptr(gyp), _ptr(ctx.wd3), None, None, None, _ptr(dx), None, B, Tout + 2 * (k - 1), Co,
                                                 Ci, k, _stream()), "ign_clconv_fwd_x6(dx)")
        if ctx.needs_input_grad[1]:
            x6 = k in (2, 3, 5, 8) and LINEAR_WGRAD == "bf16x6"
            if not x6 and k > 1 and Ci % 4 == 0 and LINEAR_WGRAD == "bf16x6":
                # tap counts without an instantiated multi-tap kernel (k = 4, 7, 11, ...): one k = 1 GEMM per tap on FLAT row
                # views.  With dy zero-padded by k-1 rows at the END of every sample, dW[:, :, j] = dy_flat[0 : M-j]^T x_flat[j : M]
                # -- the pairs that straddle two samples multiply zero rows -- so each tap is the Linear-layer weight gradient
                # on the split-bf16 kernel with the operand pointer advanced by j rows (no copies of x).
                M = B * Tin
                dye = torch.nn.functional.pad(gy, (0, 0, 0, k - 1)).contiguous()
                dwt = torch.empty(k, Co, Ci, device=gy.device, dtype=torch.float32)
                ws = torch.empty(int(L.ign_clconv_wgrad_x6_workspace_bytes(1, M, Ci, Co, 1)) // 4, device=gy.device,
                                 dtype=torch.float32)
                fn = L.ign_linear_wgrad_bf16 if ctx.bf16 else L.ign_linear_wgrad_x6
                for j in range(k):
                    xj = ctypes.c_void_p(x.data_ptr() + 4 * j * Ci)
                    if ctx.h3:
                        _lib.check(L.ign_linear_wgrad_h3(_ptr(dye), xj, _ptr(dwt[j]), None, _ptr(ws), _ptr(bg), _ptr(ctx.bx), M - j, Ci, Co,
                                                         _stream()), "ign_linear_wgrad_h3(tap)")
                    else:
                        _lib.check(fn(_ptr(dye), xj, _ptr(dwt[j]), None, _ptr(ws), M - j, Ci, Co, _stream()), "ign_linear_wgrad_x6(tap)")
                dw = dwt.permute(1, 2, 0).contiguous()
            else:
                dw = torch.empty(Co, Ci, k, device=gy.device, dtype=torch.float32)
                wsb, fn, name = ((L.ign_clconv_wgrad_x6_workspace_bytes, _gemm(L, ctx.bf16)[1], "ign_clconv_wgrad_x6") if x6 else
                                 (L.ign_clconv_wgrad_workspace_bytes, L.ign_clconv_wgrad, "ign_clconv_wgrad"))
                ws = torch.empty(max(1, int(wsb(B, Tin, Ci, Co, k)) // 4), device=gy.device, dtype=torch.float32)
                if ctx.h3 and x6:
                    _lib.check(L.ign_clconv_wgrad_h3(_ptr(gyp), k - 1, _ptr(x), None, None, _ptr(dw), _ptr(ws), _ptr(bg), _ptr(ctx.bx), B,
                                                     Tin, Ci, Co, k, _stream()), "ign_clconv_wgrad_h3")
                else:
                    _lib.check(fn(_ptr(gyp), k - 1, _ptr(x), None, None, _ptr(dw), _ptr(ws), B, Tin, Ci, Co, k, _stream()), name)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = gy.sum(dim=(0, 1))
        return dx, dw, db


def conv1d_cl(x_btc, w_oik, bias=None):
    """Valid stride-1 Conv1d over (B, T, C) channels-last input on the hand-written kernels (see ConvCLFn)."""
    return ConvCLFn.apply(x_btc, w_oik, bias)


class GiniGateFn(torch.autograd.Function):
    """(sbm_out, dnn_out) -> (mixture, eta); IGN/model/InterpGN.py:44-52."""

    @staticmethod
    def forward(ctx, sbm, dnn, gating_value):
        _need_gpu("gini_gate", sbm, dnn)
        sbm, dnn = sbm.contiguous(), dnn.contiguous()
        B, N = sbm.shape
        out = torch.empty_like(sbm)
        eta = torch.empty(B, 1, device=sbm.device, dtype=torch.float32)
        use, gv = (0, 0.0) if gating_value is None else (1, float(gating_value))
        _lib.check(_lib.lib().ign_gate_fwd(_ptr(sbm), _ptr(dnn), _ptr(out), _ptr(eta), B, N, gv, use, _stream()), "ign_gate_fwd")
        ctx.save_for_backward(sbm, dnn)
        ctx.gv = (use, gv)
        ctx.set_materialize_grads(False)              # eta usually carries no gradient: no zero fill for it
        return out, eta

    @staticmethod
    def backward(ctx, gout, geta):
        sbm, dnn = ctx.saved_tensors
        B, N = sbm.shape
        use, gv = ctx.gv
        gs, gd = torch.empty_like(sbm), torch.empty_like(dnn)
        if gout is None and geta is None:
            return None, None, None
        gout = torch.zeros_like(sbm) if gout is None else gout
        geta = geta.contiguous() if geta is not None else None
        _lib.check(_lib.lib().ign_gate_bwd(_ptr(sbm), _ptr(dnn), _ptr(gout.contiguous()), _ptr(geta), _ptr(gs), _ptr(gd), B, N,
                                           gv, use, _stream()), "ign_gate_bwd")
        return gs, gd, None


def gini_gate(sbm_out, dnn_out, gating_value=None):
    return GiniGateFn.apply(sbm_out.float(), dnn_out.float(), gating_value)


_UNIT = {}


def unit_grad(device):
    """The constant 1.0 on `device` (one cached 0-dim tensor): the root gradient of `backward(loss)`."""
    t = _UNIT.get(device)
    if t is None:
        t = _UNIT[device] = torch.ones((), device=device, dtype=torch.float32)
    return t


def backward(loss):
    """loss.backward() without the two launches autograd spends on the root of the graph: `ones_like(loss)` is replaced by a
    cached constant, and ops.ign_loss recognises that constant (by address) and hands out its saved logit gradients unscaled."""
    if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32:
        loss.backward(gradient=unit_grad(loss.device))
    else:
        loss.backward()


class IgnLossFn(torch.autograd.Function):
    """CE(gate(sbm, dnn), y) + beta * CE(sbm, y) [+ reg] with both logit gradients from one launch (ign_loss_fwd_bwd_reg)."""

    @staticmethod
    def forward(ctx, sbm, dnn, y, beta, reg):
        _need_gpu("ign_loss", sbm, dnn, reg)
        sbm, dnn = sbm.contiguous(), dnn.contiguous()
        y = y.contiguous().long()
        B, N = sbm.shape
        out = torch.empty_like(sbm)
        gsd = torch.empty(2, B, N, device=sbm.device, dtype=torch.float32)       # (d loss / d sbm, d loss / d dnn), one buffer
        eta = torch.empty(B, 1, device=sbm.device, dtype=torch.float32)
        loss2 = torch.empty(3, device=sbm.device, dtype=torch.float32)
        if reg is not None:
            reg = reg.contiguous().reshape(-1)
            if reg.numel() != 1:
                raise _lib.IgnError(f"ign_loss: the regulariser must be one value, got {tuple(reg.shape)}")
        _lib.check(_lib.lib().ign_loss_fwd_bwd_reg(_ptr(sbm), _ptr(dnn), _ptr(y), _ptr(reg), _ptr(out), _ptr(eta), _ptr(loss2),
                                                   _ptr(gsd[0]), _ptr(gsd[1]), B, N, float(beta), _stream()), "ign_loss_fwd_bwd")
        ctx.save_for_backward(gsd)
        ctx.reg_shape = None if reg is None else tuple(reg.shape)
        ctx.mark_non_differentiable(out, eta)
        ctx.set_materialize_grads(False)              # no zero-filled "gradients" of the two reporting outputs per step
        return loss2[2], out, eta

    @staticmethod
    def backward(ctx, gl, gout, geta):
        if gl is None:
            return None, None, None, None, None
        (gsd,) = ctx.saved_tensors
        unit = _UNIT.get(gl.device)
        # the root gradient of ops.backward(): exactly 1 -- no scaling launch; otherwise one launch for both logit gradients
        g = gsd if (unit is not None and gl.data_ptr() == unit.data_ptr()) else gl * gsd
        return g[0], g[1], None, None, (gl.reshape(ctx.reg_shape) if ctx.reg_shape is not None else None)


def ign_loss(sbm_out, dnn_out, y, beta=1.0, reg=None):
    """-> (CE(mix, y) + beta*CE(sbm, y) [+ reg], mix, eta); mix / eta are reporting outputs (no gradient flows through them).
    `reg`: the model's regulariser value (ModelInfo.loss, one element) -- added on the device inside the same launch, i.e. the
    whole training loss of IGN/exp/experiment_classification.py:325-329 (its gradient passes straight through)."""
    return IgnLossFn.apply(sbm_out.float(), dnn_out.float(), y, beta, reg)


class Conv1SumSqFn(torch.autograd.Function):
    """m2[f] = sum_{rows,t} ((w1[f] (*) x_row)[t] - mu[f])^2 without storing the convolution (ign_conv1_sumsq_*).
    d m2 / d mu = -2 sum (y1 - mu) = 0 when mu is the batch mean, which is the only use (BatchNorm-1 of EEG-CNN)."""

    @staticmethod
    def forward(ctx, x_rows, w1, mu, pad_left):
        _need_gpu("conv1_sumsq", x_rows, w1, mu)
        x_rows, w1, mu = x_rows.contiguous(), w1.contiguous(), mu.contiguous()
        R, T = x_rows.shape
        F1, k1 = w1.shape
        L = _lib.lib()
        ws = torch.empty(L.ign_conv1_sumsq_workspace_bytes(R, F1, k1) // 4, device=x_rows.device, dtype=torch.float32)
        m2 = torch.empty(F1, device=x_rows.device, dtype=torch.float32)
        _lib.check(L.ign_conv1_sumsq_fwd(_ptr(x_rows), _ptr(w1), _ptr(mu), _ptr(m2), _ptr(ws), R, T, F1, k1, int(pad_left),
                                         _stream()), "ign_conv1_sumsq_fwd")
        ctx.save_for_backward(x_rows, w1, mu)
        ctx.pl = int(pad_left)
        return m2

    @staticmethod
    def backward(ctx, g):
        x_rows, w1, mu = ctx.saved_tensors
        R, T = x_rows.shape
        F1, k1 = w1.shape
        L = _lib.lib()
        ws = torch.empty(L.ign_conv1_sumsq_workspace_bytes(R, F1, k1) // 4, device=x_rows.device, dtype=torch.float32)
        G = torch.empty(F1, k1, device=x_rows.device, dtype=torch.float32)
        _lib.check(L.ign_conv1_sumsq_bwd(_ptr(x_rows), _ptr(w1), _ptr(mu), _ptr(G), _ptr(ws), R, T, F1, k1, ctx.pl, _stream()),
                   "ign_conv1_sumsq_bwd")
        return None, 2.0 * g.unsqueeze(1) * G, None, None


def autocorr(x_rows, K):
    """C[d] = sum over rows and u of x[row,u] * x[row,u+d], d < K <= 128, as float64 (ign_autocorr_fwd; no gradient: the
    operand is input data)."""
    _need_gpu("autocorr", x_rows)
    L = _lib.lib()
    x_rows = x_rows.contiguous()
    R, T = x_rows.shape
    part = torch.empty(int(L.ign_autocorr_parts(R)), K, device=x_rows.device, dtype=torch.float32)
    _lib.check(L.ign_autocorr_fwd(_ptr(x_rows), _ptr(part), R, T, K, _stream()), "ign_autocorr_fwd")
    return part.sum(dim=0, dtype=torch.float64)


def edge_lagprod(x_rows, k, pad_left):
    """(Dh, Dt), float64 (k-1, k): Dh[s, d] = sum_rows xp[s] xp[s+d] over the first k-1 samples of the zero-padded rows (0 where
    s + d >= k-1), Dt the same over the k-1 samples from position T on (ign_edge_lagprod_fwd; no gradient: input data)."""
    _need_gpu("edge_lagprod", x_rows)
    L = _lib.lib()
    x_rows = x_rows.contiguous()
    R, T = x_rows.shape
    part = torch.empty(int(L.ign_edge_lagprod_parts(R)), 2, 124, 128, device=x_rows.device, dtype=torch.float32)
    _lib.check(L.ign_edge_lagprod_fwd(_ptr(x_rows), _ptr(part), R, T, int(k), int(pad_left), _stream()), "ign_edge_lagprod_fwd")
    tot = part[:, :, :k - 1, :k].sum(dim=0, dtype=torch.float64)
    return tot[0], tot[1]


def conv1_sumsq(x_rows, w1, mu, pad_left):
    return Conv1SumSqFn.apply(x_rows, w1, mu, pad_left)


def bn1_data_stats(x_rows, k, pad_left):
    """(G, S), float64: the window Gram matrix (k, k) and the per-tap sums (k) of the zero-padded rows of `x_rows` (R, T) -- the
    data side of BatchNorm-1's batch statistics (models/eegcnn.py), one C call = four launches (ign_bn1_data_stats: lag sums and
    row sums, edge terms and column sums, float64 sums of the per-block partials, assembly).  No gradient: the operand is input
    data.  2 <= k <= 125, k <= T <= 1024."""
    _need_gpu("bn1_data_stats", x_rows)
    L = _lib.lib()
    x_rows = x_rows.contiguous()
    R, T = x_rows.shape
    dev = x_rows.device
    ws = torch.empty(int(L.ign_bn1_data_stats_workspace_bytes(R, T, int(k))), device=dev, dtype=torch.uint8)
    G = torch.empty(k, k, device=dev, dtype=torch.float64)
    S = torch.empty(k, device=dev, dtype=torch.float64)
    _lib.check(L.ign_bn1_data_stats(_ptr(x_rows), R, T, int(k), int(pad_left), _ptr(ws), _ptr(G), _ptr(S), _stream()),
               "ign_bn1_data_stats")
    return G, S


class Bn1FoldFn(torch.autograd.Function):
    """(w1 (F1,k), gamma, beta (F1), rs (F1*D)) -> (alpha, cshift) (F1*D each): BatchNorm-1 with batch statistics folded into the
    per-channel affine map the fused BatchNorm-2 op absorbs -- mean and variance are the linear / quadratic form of the filter
    over the data statistics (S, G), so forward and backward are closed-form in the parameters (ign_bn1_fold_fwd / _bwd).
    Running statistics (nullable) are updated by the forward launch."""

    @staticmethod
    def forward(ctx, w1, gamma, beta, rs, G, S, n, eps, momentum, run_mean, run_var, Dm):
        _need_gpu("bn1_fold", w1, gamma, beta, rs)
        L = _lib.lib()
        w1, gamma, beta, rs = w1.contiguous(), gamma.contiguous(), beta.contiguous(), rs.contiguous()
        F1, k = w1.shape
        alpha = torch.empty(F1 * Dm, device=w1.device, dtype=torch.float32)
        cshift = torch.empty_like(alpha)
        saved = torch.empty(F1, k + 4, device=w1.device, dtype=torch.float64)
        _lib.check(L.ign_bn1_fold_fwd(_ptr(w1), _ptr(gamma), _ptr(beta), _ptr(rs), _ptr(G), _ptr(S), float(n), float(eps),
                                      float(momentum), _ptr(run_mean), _ptr(run_var), _ptr(alpha), _ptr(cshift), _ptr(saved), F1, k,
                                      int(Dm), _stream()), "ign_bn1_fold_fwd")
        ctx.save_for_backward(w1, gamma, rs, S, saved)
        ctx.meta = (float(n), int(Dm))
        return alpha, cshift

    @staticmethod
    def backward(ctx, g_alpha, g_cshift):
        w1, gamma, rs, S, saved = ctx.saved_tensors
        n, Dm = ctx.meta
        F1, k = w1.shape
        g_alpha, g_cshift = g_alpha.contiguous(), g_cshift.contiguous()
        g_w1 = torch.empty_like(w1)
        g_gamma = torch.empty_like(gamma)
        g_beta = torch.empty_like(gamma)
        g_rs = torch.empty_like(rs)
        _lib.check(_lib.lib().ign_bn1_fold_bwd(_ptr(g_alpha), _ptr(g_cshift), _ptr(w1), _ptr(gamma), _ptr(rs), _ptr(S), _ptr(saved), n,
                                               _ptr(g_w1), _ptr(g_gamma), _ptr(g_beta), _ptr(g_rs), F1, k, Dm, _stream()),
                   "ign_bn1_fold_bwd")
        return g_w1, g_gamma, g_beta, g_rs, None, None, None, None, None, None, None, None


def bn1_fold(w1, gamma, beta, rs, G, S, n, eps, momentum, running_mean, running_var, Dm):
    return Bn1FoldFn.apply(w1, gamma, beta, rs, G, S, n, eps, momentum, running_mean, running_var, Dm)


class DwConv1dFn(torch.autograd.Function):
    """Depthwise 'same' 1-D convolution y[b,c,t] = sum_j w[c,j] xpad[b,c,t+j]  (ign_dwconv1d_*)."""

    @staticmethod
    def forward(ctx, x, w, pad_left):
        _need_gpu("dwconv1d", x, w)
        x, w = x.contiguous(), w.contiguous()
        B, C, T = x.shape
        k = w.shape[1]
        y = torch.empty_like(x)
        _lib.check(_lib.lib().ign_dwconv1d_fwd(_ptr(x), _ptr(w), _ptr(y), B, C, T, k, int(pad_left), 0, _stream()),
                   "ign_dwconv1d_fwd")
        ctx.save_for_backward(x, w)
        ctx.pl = int(pad_left)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        B, C, T = x.shape
        k = w.shape[1]
        gy = gy.contiguous()
        L = _lib.lib()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(L.ign_dwconv1d_fwd(_ptr(gy), _ptr(w), _ptr(gx), B, C, T, k, k - 1 - ctx.pl, 1, _stream()),
                       "ign_dwconv1d_fwd(flip)")
        if ctx.needs_input_grad[1]:
            ws = torch.empty(L.ign_dwconv1d_bwd_weight_workspace_bytes(B, C, k) // 4, device=x.device, dtype=torch.float32)
            gw = torch.empty_like(w)
            _lib.check(L.ign_dwconv1d_bwd_weight(_ptr(x), _ptr(gy), _ptr(gw), _ptr(ws), B, C, T, k, ctx.pl, _stream()),
                       "ign_dwconv1d_bwd_weight")
        return gx, gw, None


def dwconv1d(x, w, pad_left):
    return DwConv1dFn.apply(x, w, pad_left)


class ChanContractFn(torch.autograd.Function):
    """u[b,o,t] = sum_c W[o,c] x[b,c,t]  (ign_chan_contract_*): the EEG-CNN's electrode contraction and pointwise convolution
    (IGN/model/eegcnn.py:71,79) on (B, channels, T) tensors.  Co <= 64, Ci <= 128."""

    @staticmethod
    def _pad64(w_t):                       # (Ci, Co) -> (Ci, 64), zero columns
        return w_t.contiguous() if w_t.shape[1] == 64 else torch.nn.functional.pad(w_t, (0, 64 - w_t.shape[1])).contiguous()

    @staticmethod
    def _run(x, w_oc):
        B, Ci, T = x.shape
        Co = w_oc.shape[0]
        u = torch.empty(B, Co, T, device=x.device, dtype=torch.float32)
        wt = ChanContractFn._pad64(w_oc.t())
        _lib.check(_lib.lib().ign_chan_contract_fwd(_ptr(x), _ptr(wt), _ptr(u), B, Ci, Co, T, _stream()), "ign_chan_contract_fwd")
        return u

    @staticmethod
    def forward(ctx, x, w):
        _need_gpu("chan_contract", x, w)
        if w.shape[0] > 64 or w.shape[1] > 128 or w.shape[1] != x.shape[1]:
            raise _lib.IgnError(f"chan_contract: W {tuple(w.shape)} on x {tuple(x.shape)} (Co <= 64, Ci <= 128)")
        x, w = x.contiguous(), w.contiguous()
        ctx.save_for_backward(x, w)
        return ChanContractFn._run(x, w)

    @staticmethod
    def backward(ctx, gu):
        x, w = ctx.saved_tensors
        gu = gu.contiguous()
        B, Ci, T = x.shape
        Co = w.shape[0]
        L = _lib.lib()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if Ci > 64:
                raise _lib.IgnError("chan_contract: input gradient needs Ci <= 64 (the electrode axis is data and takes none)")
            gx = ChanContractFn._run(gu, w.t().contiguous())
        if ctx.needs_input_grad[1]:
            ws = torch.empty(L.ign_chan_contract_bwd_weight_workspace_bytes(B, Ci, Co, T) // 4, device=x.device, dtype=torch.float32)
            gw = torch.empty_like(w)
            _lib.check(L.ign_chan_contract_bwd_weight(_ptr(gu), _ptr(x), _ptr(gw), _ptr(ws), B, Ci, Co, T, _stream()),
                       "ign_chan_contract_bwd_weight")
        return gx, gw


def chan_contract(x_bct, w_oc):
    return ChanContractFn.apply(x_bct, w_oc)


class BnEluPoolFn(torch.autograd.Function):
    """AvgPool_P(ELU(BatchNorm_train(alpha * v + c))) for v (B, C, T) -- BatchNorm2d + ELU + AvgPool2d((1,P)) of the EEG-CNN
    block (IGN/model/eegcnn.py:72-74,80-82) as one op over the hand-written kernels, with the per-channel affine map in FRONT
    of the BatchNorm (block 1: BatchNorm-1 folded behind the electrode contraction) absorbed analytically:

        y = alpha v + c;  var_y = alpha^2 var_v;  r = rsqrt(var_y + eps);  z = gamma alpha r (v - mean_v) + beta
        (c cancels in the normalisation: its gradient is exactly 0 -- returned as None; it only enters the running mean)

    backward (dz = ELU'(z) dout / P;  S1 = sum dz,  S2 = sum dz (v - mean_v);  n = B T):
        dgamma = alpha r S2,  dbeta = S1,  dv = gamma alpha r (dz - S1/n - alpha^2 r^2 (v - mean_v) S2 / n),
        dalpha = gamma r^3 eps S2          (the only path: BatchNorm is scale-invariant up to eps)
    The per-channel algebra runs in ign_bn_fold_fwd / _bwd (one launch each, float64), which also applies the running-statistics
    update (momentum, unbiased variance) of nn.BatchNorm2d in place when `run_mean` / `run_var` are given."""

    @staticmethod
    def forward(ctx, v, alpha, cshift, gamma, beta, P, eps, run_mean, run_var, momentum):
        _need_gpu("bn_elu_pool", v, gamma, beta)
        v = v.contiguous()
        B, C, T = v.shape
        L = _lib.lib()
        dev = v.device
        f32 = lambda t: None if t is None else t.detach().float().contiguous()
        alpha32, cshift32, gamma32, beta32 = f32(alpha), f32(cshift), f32(gamma), f32(beta)
        sums = torch.empty(C, 2, device=dev, dtype=torch.float64)
        ws = torch.empty(L.ign_chan_stats_workspace_bytes(B, C) // 8, device=dev, dtype=torch.float64)
        _lib.check(L.ign_chan_stats(_ptr(v), _ptr(sums), _ptr(ws), B, C, T, _stream()), "ign_chan_stats")
        scale32 = torch.empty(C, device=dev, dtype=torch.float32)
        shift32 = torch.empty(C, device=dev, dtype=torch.float32)
        fold = torch.empty(C, 2, device=dev, dtype=torch.float64)            # (mean_v, r)
        _lib.check(L.ign_bn_fold_fwd(_ptr(sums), _ptr(alpha32), _ptr(cshift32), _ptr(gamma32), _ptr(beta32), _ptr(scale32),
                                     _ptr(shift32), _ptr(fold), _ptr(run_mean), _ptr(run_var), C, B * T, float(eps),
                                     float(momentum if momentum is not None else 0.0), _stream()), "ign_bn_fold_fwd")
        out = torch.empty(B, C, T // P, device=dev, dtype=torch.float32)
        _lib.check(L.ign_affine_elu_pool_fwd(_ptr(v), _ptr(scale32), _ptr(shift32), _ptr(out), B, C, T, int(P), _stream()),
                   "ign_affine_elu_pool_fwd")
        ctx.save_for_backward(v, scale32, shift32, fold, alpha32, gamma32)
        ctx.P, ctx.eps = int(P), float(eps)
        return out

    @staticmethod
    def backward(ctx, gout):
        v, scale32, shift32, fold, alpha32, gamma32 = ctx.saved_tensors
        gout = gout.contiguous()
        B, C, T = v.shape
        L = _lib.lib()
        dev = v.device
        mean32 = fold[:, 0].float().contiguous()
        sums = torch.empty(C, 2, device=dev, dtype=torch.float64)
        ws = torch.empty(L.ign_chan_stats_workspace_bytes(B, C) // 8, device=dev, dtype=torch.float64)
        _lib.check(L.ign_bn_elu_pool_bwd_sums(_ptr(v), _ptr(gout), _ptr(scale32), _ptr(shift32), _ptr(mean32), _ptr(sums), _ptr(ws),
                                              B, C, T, ctx.P, _stream()), "ign_bn_elu_pool_bwd_sums")
        coef = torch.empty(6, C, device=dev, dtype=torch.float32)            # ka, kb, kc, dgamma, dbeta, dalpha
        _lib.check(L.ign_bn_fold_bwd(_ptr(sums), _ptr(fold), _ptr(alpha32), _ptr(gamma32), _ptr(coef[0]), _ptr(coef[1]), _ptr(coef[2]),
                                     _ptr(coef[3]), _ptr(coef[4]), _ptr(coef[5]) if alpha32 is not None else None, C, B * T, ctx.eps,
                                     _stream()), "ign_bn_fold_bwd")
        dv = None
        if ctx.needs_input_grad[0]:
            dv = torch.empty_like(v)
            _lib.check(L.ign_bn_elu_pool_bwd_apply(_ptr(v), _ptr(gout), _ptr(scale32), _ptr(shift32), _ptr(coef[0]), _ptr(coef[1]),
                                                   _ptr(coef[2]), _ptr(dv), B, C, T, ctx.P, _stream()), "ign_bn_elu_pool_bwd_apply")
        return dv, (coef[5] if alpha32 is not None else None), None, coef[3], coef[4], None, None, None, None, None


def bn_elu_pool(v, bn, P, alpha=None, cshift=None):
    """BatchNorm2d module `bn` (its weight / bias / running statistics / momentum / eps) + ELU + AvgPool((1,P)) on v (B,C,T);
    `alpha`, `cshift`: per-channel affine map y = alpha v + cshift applied in front of the BatchNorm (see BnEluPoolFn).
    Training mode uses batch statistics and updates the running ones like nn.BatchNorm2d; eval mode uses the running ones."""
    _need_gpu("bn_elu_pool", v)
    C = v.shape[1]
    # affine=False: the identity affine map (constants, no gradient)
    gamma = bn.weight if bn.weight is not None else torch.ones(C, device=v.device, dtype=torch.float32)
    beta = bn.bias if bn.bias is not None else torch.zeros(C, device=v.device, dtype=torch.float32)
    if bn.training or not bn.track_running_stats:
        track = bn.training and bn.track_running_stats
        momentum = bn.momentum
        if track:
            bn.num_batches_tracked.add_(1)
            if momentum is None:      # nn.BatchNorm with momentum=None: cumulative moving average, factor 1 / num_batches_tracked
                momentum = 1.0 / float(bn.num_batches_tracked)
        return BnEluPoolFn.apply(v, alpha, cshift, gamma, beta, P, bn.eps, bn.running_mean if track else None,
                                 bn.running_var if track else None, momentum)
    # eval: z = gamma (alpha v + c - running_mean) / sqrt(running_var + eps) + beta -- an affine map, then the apply kernel
    r = torch.rsqrt(bn.running_var + bn.eps)
    a = alpha if alpha is not None else torch.ones_like(r)
    c = cshift if cshift is not None else torch.zeros_like(r)
    scale = (gamma * a * r).contiguous()
    shift = (gamma * (c - bn.running_mean) * r + beta).contiguous()
    return AffineEluPoolFn.apply(v, scale, shift, P)


class AffineEluPoolFn(torch.autograd.Function):
    """AvgPool_P(ELU(scale[c] v + shift[c])) with gradients to v, scale and shift (the eval-mode form of bn_elu_pool)."""

    @staticmethod
    def forward(ctx, v, scale, shift, P):
        _need_gpu("affine_elu_pool", v, scale, shift)
        v, scale, shift = v.contiguous(), scale.contiguous(), shift.contiguous()
        B, C, T = v.shape
        out = torch.empty(B, C, T // P, device=v.device, dtype=torch.float32)
        _lib.check(_lib.lib().ign_affine_elu_pool_fwd(_ptr(v), _ptr(scale), _ptr(shift), _ptr(out), B, C, T, int(P), _stream()),
                   "ign_affine_elu_pool_fwd")
        ctx.save_for_backward(v, scale, shift)
        ctx.P = int(P)
        return out

    @staticmethod
    def backward(ctx, gout):
        v, scale, shift = ctx.saved_tensors
        gout = gout.contiguous()
        B, C, T = v.shape
        L = _lib.lib()
        zero = torch.zeros_like(scale)
        sums = torch.empty(C, 2, device=v.device, dtype=torch.float64)
        ws = torch.empty(L.ign_chan_stats_workspace_bytes(B, C) // 8, device=v.device, dtype=torch.float64)
        _lib.check(L.ign_bn_elu_pool_bwd_sums(_ptr(v), _ptr(gout), _ptr(scale), _ptr(shift), _ptr(zero), _ptr(sums), _ptr(ws),
                                              B, C, T, ctx.P, _stream()), "ign_bn_elu_pool_bwd_sums")
        dv = None
        if ctx.needs_input_grad[0]:
            dv = torch.empty_like(v)
            _lib.check(L.ign_bn_elu_pool_bwd_apply(_ptr(v), _ptr(gout), _ptr(scale), _ptr(shift), _ptr(scale), _ptr(zero), _ptr(zero),
                                                   _ptr(dv), B, C, T, ctx.P, _stream()), "ign_bn_elu_pool_bwd_apply")
        return dv, sums[:, 1].float(), sums[:, 0].float(), None


class DiversityFn(torch.autograd.Function):
    """mean_{c,i,j} exp(-||w_i - w_j + 1e-6||) (1 - delta_ij) of one shapelet group (IGN/model/Shapelet.py:223-230);
    the kernel produces the loss and its gradient together, backward only scales the saved gradient."""

    @staticmethod
    def forward(ctx, w):
        _need_gpu("diversity", w)
        w = w.contiguous()
        K, C, L = w.shape
        part = torch.empty(C, device=w.device, dtype=torch.float32)
        gw = torch.empty_like(w)
        _lib.check(_lib.lib().ign_diversity_fwd_bwd(_ptr(w), _ptr(part), _ptr(gw), K, C, L, 1e-6, _stream()),
                   "ign_diversity_fwd_bwd")
        ctx.save_for_backward(gw)
        return part.sum()

    @staticmethod
    def backward(ctx, g):
        (gw,) = ctx.saved_tensors
        return gw * g


def diversity(w):
    if not w.is_cuda or w.dtype != torch.float32 or w.shape[0] > 16:
        return None
    return DiversityFn.apply(w)

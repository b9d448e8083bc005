"""The FCN expert's convolution stack as ONE autograd node over the channels-last implicit-GEMM kernels
(include/ign_abi.h: ign_clconv_*, ign_bn_*; csrc/ign_clconv_{f32,x6}.hip, ign_bn.hip).

Replaces ``block1 -> block2 -> block3 -> AdaptiveAvgPool1d`` of IGN/model/FullyConvNet.py:31-57.  What reaches HBM per
block is only the raw convolution output y_l (needed by the backward anyway): BatchNorm's statistics come out of the
GEMM epilogue, its affine + ReLU are applied while the NEXT GEMM stages its operand, and the last block's
BatchNorm + ReLU + average pool is one reduction pass.  Backward mirrors it: the data-gradient GEMM applies the ReLU
mask of the block below and emits that block's BatchNorm-backward sums; one elementwise pass per block turns them
into dL/dy (written zero-padded, the layout the GEMMs read); the weight-gradient GEMM recomputes relu(bn(y)) on the
fly.  torch supplies memory and the autograd plumbing only.  There is no CPU path.
"""
import ctypes
import os

import torch

from . import _lib


# Arithmetic of the forward / data-gradient GEMMs (the weight gradient is fp32 MFMA in both):
#   "bf16x6": exact 3-way bf16 split of every fp32 operand, six partial products accumulated in fp32 on the bf16 matrix
#             cores -- error vs float64 at the fp32 kernel's level (<= 3e-6 of max; tests/test_gpu_fcn.py), 1.6x its speed;
#   "f32":    v_mfma_f32_32x32x2_f32 (runs at the fp32 VECTOR rate on gfx950);
#   "bf16":   operands rounded to bf16, ONE product, fp32 accumulation -- what torch.autocast(bfloat16) computes for a
#             convolution; selected automatically inside an autocast region (the reference's default mode, `--amp` not passed).
#   "f16x3":  every operand tensor scaled by a power of two (from a device-side bound of its magnitude), split into TWO fp16
#             terms, THREE partial products accumulated in fp32 on the fp16 matrix cores -- the same fp32-level accuracy against
#             float64 as "bf16x6" with half the matrix instructions and two operand planes instead of three (1.5-1.6x its speed;
#             include/ign_abi.h, "h3").  The default in training mode; blocks that normalise with running statistics (eval mode)
#             have no hard bound on their activations and use "bf16x6".
CONV_MATH = os.environ.get("IGN_CONV_MATH", "f16x3")
DEBUG = None       # set to a dict to capture the backward intermediates (tests/diag_fcn.py)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_stream = _lib.stream


def _attached_bound(t):
    """ops.cached_bound without importing ops (which imports this module)."""
    cb = getattr(t, "_ign_bound", None)
    if cb is None or cb[1] != t._version or cb[2] != t.data_ptr():
        return None
    if torch.cuda.is_current_stream_capturing() and not (len(cb) > 3 and cb[3]):
        return None
    return cb[0]


class BnState:
    """Non-tensor carrier for one block's BatchNorm1d buffers and mode (torch.nn.BatchNorm semantics: batch statistics
    when training or when no running statistics are tracked; the running buffers are updated in place by
    ign_bn_finalize_fwd only in training mode)."""

    def __init__(self, bn):
        self.eps = float(bn.eps)
        self.use_batch_stats = bn.training or not bn.track_running_stats
        self.running_mean = self.running_var = None
        self.momentum = 0.0
        self.bump = None            # the num_batches_tracked counter to increment (fcn_body bumps every block's in ONE launch)
        if bn.track_running_stats:
            self.running_mean, self.running_var = bn.running_mean, bn.running_var
            if bn.training:
                if bn.momentum is None:     # cumulative moving average: the factor needs the counter's value on the host
                    if bn.num_batches_tracked is not None:
                        bn.num_batches_tracked.add_(1)
                    self.momentum = 1.0 / float(bn.num_batches_tracked)
                else:
                    self.bump = bn.num_batches_tracked
                    self.momentum = float(bn.momentum)
        self.update_running = bn.training and bn.track_running_stats


class FcnBodyFn(torch.autograd.Function):
    """forward(x (B,T,C), states, w1, b1, gamma1, beta1, w2, ..., beta3) -> pooled (B, C3)."""

    @staticmethod
    def forward(ctx, x, states_math, *params):
        L = _lib.lib()
        states, math, has_head = states_math
        head_w, head_b = (params[-2], params[-1]) if has_head else (None, None)
        if has_head:
            params = params[:-2]
        if not x.is_cuda or x.dtype != torch.float32:
            raise _lib.IgnError(f"fcn_body: needs a float32 GPU tensor, got {x.dtype} on {x.device} (no CPU fallback)")
        x = x.contiguous()
        B = x.shape[0]
        nl = len(params) // 4
        dev = x.device
        f32 = dict(device=dev, dtype=torch.float32)
        need_grad = any(ctx.needs_input_grad[2:2 + len(params)])
        if math == "f16x3" and not (all(st.use_batch_stats for st in states) and nl <= 8):
            math = "bf16x6"           # running statistics: no hard bound on the normalised activations
        h3 = math == "f16x3"
        x6 = math in ("bf16x6", "bf16", "f16x3")
        conv_fwd = L.ign_clconv_fwd_bf16 if math == "bf16" else L.ign_clconv_fwd_x6
        slots = zbuf = xbound = None
        fp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 4 * off)        # address of float `off` of a float32 tensor
        inputs, affine, wds, shapes = [x], [], [], []
        pa = pb = None
        packed = None
        if x6 and nl <= 8:
            # prologue: the weights of every block packed (split into bf16 planes) in ONE launch, which also bumps the
            # BatchNorm step counters of the blocks that track running statistics
            ws_ = [params[4 * l].contiguous() for l in range(nl)]
            wts = [torch.empty(int(L.ign_clconv_x3_elems(w.shape[0], w.shape[1], w.shape[2])), device=dev, dtype=torch.bfloat16)
                   for w in ws_]
            wds_ = [torch.empty(int(L.ign_clconv_x3_elems(w.shape[1], w.shape[0], w.shape[2])), device=dev, dtype=torch.bfloat16)
                    if (l > 0 and need_grad) else None for l, w in enumerate(ws_)]
            vpa, ia, lla = ctypes.c_void_p * nl, ctypes.c_int * nl, ctypes.c_longlong * nl
            pv = lambda ts: vpa(*[(t.data_ptr() if t is not None else None) for t in ts])
            bumps = [st.bump for st in states]
            bump_tab = pv(bumps) if any(b is not None for b in bumps) else None
            dims = (ia(*[w.shape[0] for w in ws_]), ia(*[w.shape[1] for w in ws_]), ia(*[w.shape[2] for w in ws_]))
            if h3:
                # operand bounds (4 floats per block: |W|, |block input|, |dL/dy|, spare): weights and the BatchNorm-derived
                # activation bounds from the parameters in one launch, the raw input's from one pass over x
                slots = torch.empty(4 * nl, **f32)
                gam = [None] + [params[4 * (l - 1) + 2].contiguous() for l in range(1, nl)]
                bet = [None] + [params[4 * (l - 1) + 3].contiguous() for l in range(1, nl)]
                Touts, T_ = [], x.shape[1]
                for w in ws_:
                    T_ = T_ - w.shape[2] + 1
                    Touts.append(T_)
                # (the same launch clears the buffer the backward returns as the convolution biases' gradients: a bias in front of
                # a batch-statistics BatchNorm has an identically zero gradient -- the batch mean removes it)
                zbuf = torch.empty(sum(w.shape[0] for w in ws_), **f32) if need_grad else None
                _lib.check(L.ign_fcn_scan(nl, pv(ws_), lla(*[w.numel() for w in ws_]), pv(gam), pv(bet),
                                          ia(*([0] + [ws_[l - 1].shape[0] for l in range(1, nl)])),
                                          lla(*([0] + [B * Touts[l - 1] for l in range(1, nl)])), _ptr(slots), _ptr(zbuf),
                                          zbuf.numel() if zbuf is not None else 0, _stream()), "ign_fcn_scan")
                # max |x|: attached to the tensor by a producer that had the batch in its hands (ops.instance_norm when both
                # experts share a stream), else one pass over x
                xbound = _attached_bound(x)
                if xbound is None:
                    _lib.check(L.ign_absmax(_ptr(x), x.numel(), fp(slots, 1), _stream()), "ign_absmax")
                    xbound = slots[1:2]
                _lib.check(L.ign_clconv_pack_weights_h2_multi(nl, pv(ws_), pv(wts), pv(wds_), *dims, bump_tab,
                                                              vpa(*[slots.data_ptr() + 16 * l for l in range(nl)]), _stream()),
                           "ign_clconv_pack_weights_h2_multi")
            else:
                _lib.check(L.ign_clconv_pack_weights_x3_multi(nl, pv(ws_), pv(wts), pv(wds_), *dims, bump_tab, _stream()),
                           "ign_clconv_pack_weights_x3_multi")
            packed = (wts, wds_)
        else:
            bumps = [st.bump for st in states if st.bump is not None]
            if bumps:
                torch._foreach_add_(bumps, 1)
        for l in range(nl):
            w, b, gamma, beta = (p.contiguous() for p in params[4 * l:4 * l + 4])
            Co, Ci, k = w.shape
            h = inputs[-1]
            Tin = h.shape[1]
            Tout = Tin - k + 1
            if h.shape[2] != Ci or Tout <= 0 or Co % 4:
                raise _lib.IgnError(f"fcn_body block {l + 1}: input {tuple(h.shape)} vs weight {tuple(w.shape)}")
            st = states[l]
            y = torch.empty(B, Tout, Co, **f32)
            nparts = int(L.ign_clconv_x6_mtiles(B, Tout) if x6 else L.ign_clconv_mtiles(B * Tout))
            part = torch.empty(nparts, 2, Co, **f32) if st.use_batch_stats else None
            want_wd = l > 0 and need_grad
            if x6:
                if packed is not None:
                    wt, wd = packed[0][l], packed[1][l]
                else:
                    wt = torch.empty(int(L.ign_clconv_x3_elems(Co, Ci, k)), device=dev, dtype=torch.bfloat16)
                    wd = torch.empty(int(L.ign_clconv_x3_elems(Ci, Co, k)), device=dev, dtype=torch.bfloat16) if want_wd else None
                    _lib.check(L.ign_clconv_pack_weights_x3(_ptr(w), _ptr(wt), _ptr(wd), Co, Ci, k, _stream()),
                               "ign_clconv_pack_weights_x3")
                if h3:
                    _lib.check(L.ign_clconv_fwd_h3(_ptr(h), _ptr(wt), _ptr(b), _ptr(pa), _ptr(pb), _ptr(y), _ptr(part),
                                                   fp(slots, 4 * l + 1) if l else _ptr(xbound), fp(slots, 4 * l), B, Tin, Ci, Co, k,
                                                   _stream()), "ign_clconv_fwd_h3")
                else:
                    _lib.check(conv_fwd(_ptr(h), _ptr(wt), _ptr(b), _ptr(pa), _ptr(pb), _ptr(y), _ptr(part), B, Tin, Ci, Co, k,
                                        _stream()), "ign_clconv_fwd_x6")
            else:
                wt = torch.empty(Co, k * Ci, **f32)
                wd = torch.empty(Ci, k * Co, **f32) if want_wd else None
                _lib.check(L.ign_clconv_pack_weights(_ptr(w), _ptr(wt), _ptr(wd), Co, Ci, k, _stream()), "ign_clconv_pack_weights")
                _lib.check(L.ign_clconv_fwd(_ptr(h), _ptr(wt), _ptr(b), _ptr(pa), _ptr(pb), _ptr(y), _ptr(part), B, Tin, Ci, Co, k,
                                            _stream()), "ign_clconv_fwd")
            a, bb, mean, invstd = (torch.empty(Co, **f32) for _ in range(4))
            if st.use_batch_stats:
                _lib.check(L.ign_bn_finalize_fwd(_ptr(part), nparts, B * Tout, Co, _ptr(gamma), _ptr(beta), st.eps, st.momentum,
                                                 _ptr(st.running_mean if st.update_running else None),
                                                 _ptr(st.running_var if st.update_running else None), _ptr(a), _ptr(bb), _ptr(mean),
                                                 _ptr(invstd), _stream()), "ign_bn_finalize_fwd")
            else:
                _lib.check(L.ign_bn_affine_eval(_ptr(st.running_mean), _ptr(st.running_var), _ptr(gamma), _ptr(beta), st.eps, Co,
                                                _ptr(a), _ptr(bb), _ptr(mean), _ptr(invstd), _stream()), "ign_bn_affine_eval")
            inputs.append(y)
            affine.append((a, bb, mean, invstd))
            wds.append(wd)
            shapes.append((Co, Ci, k, Tin, Tout))
            pa, pb = a, bb
        y_last = inputs[-1]
        Cl, Tl = y_last.shape[2], y_last.shape[1]
        pooled = torch.empty(B, Cl, **f32)
        head = None
        if has_head:
            # the class head in the pooling launch (a sample's pooled row is complete inside its block)
            head_w = head_w.contiguous()
            N = head_w.shape[0]
            if head_w.shape[1] != Cl or head_w.dtype != torch.float32:
                raise _lib.IgnError(f"fcn_body: head weight {tuple(head_w.shape)} {head_w.dtype} does not fit {Cl} pooled channels")
            out = torch.empty(B, N, **f32)
            _lib.check(L.ign_bn_relu_pool_head_fwd(_ptr(y_last), _ptr(pa), _ptr(pb), _ptr(pooled), _ptr(head_w), _ptr(head_b), _ptr(out),
                                                   B, Tl, Cl, N, _stream()), "ign_bn_relu_pool_head_fwd")
            head = (pooled, head_w, head_b is not None)
        else:
            out = pooled
            _lib.check(L.ign_bn_relu_pool_fwd(_ptr(y_last), _ptr(pa), _ptr(pb), _ptr(pooled), B, Tl, Cl, _stream()),
                       "ign_bn_relu_pool_fwd")
        ctx.head = head
        ctx.head_needs = tuple(ctx.needs_input_grad[-2:]) if has_head else None
        ctx.saved = (inputs, affine, wds, shapes, [st.use_batch_stats for st in states], math, slots, zbuf, xbound) \
            if (need_grad or (has_head and any(ctx.head_needs))) else None
        ctx.body_grad = need_grad
        return out

    @staticmethod
    def backward(ctx, gpool):
        if ctx.saved is None:
            raise _lib.IgnError("fcn_body backward called but the forward ran without parameter gradients enabled")
        if ctx.needs_input_grad[0]:
            raise _lib.IgnError("fcn_body: gradient w.r.t. the input series is not implemented (inputs are data)")
        L = _lib.lib()
        inputs, affine, wds, shapes, batch_stats, math, slots, zbuf, xbound = ctx.saved
        h3 = math == "f16x3"
        x6 = math in ("bf16x6", "bf16", "f16x3")
        fp = lambda t, off: ctypes.c_void_p(t.data_ptr() + 4 * off)
        nl = len(shapes)
        B = inputs[0].shape[0]
        dev = gpool.device
        f32 = dict(device=dev, dtype=torch.float32)
        gpool = gpool.contiguous()
        head_grads = ()
        if ctx.head is not None:
            # gpool is the gradient of the LOGITS here: head backward first (pooled-row gradient, weight and bias gradients, one launch)
            pooled, head_w, has_b = ctx.head
            N, Cl = head_w.shape
            glog = gpool
            gpool = torch.empty(B, Cl, **f32)
            ghw = torch.empty_like(head_w)
            ghb = torch.empty(N, **f32) if has_b else None
            _lib.check(L.ign_head_bwd(_ptr(glog), _ptr(pooled), _ptr(head_w), _ptr(gpool), _ptr(ghw), _ptr(ghb), B, Cl, N, Cl, _stream()),
                       "ign_head_bwd")
            head_grads = (ghw if ctx.head_needs[0] else None, ghb if ctx.head_needs[1] else None)
            if not ctx.body_grad:
                ctx.saved = None
                return (None, None, *([None] * (4 * nl)), *head_grads)
        grads = [None] * (4 * nl)
        # The bias in front of a batch-statistics BatchNorm has an identically zero gradient (the batch mean removes it): the
        # zeros of all blocks are views of ONE buffer (one fill launch)
        zoff, ztot = [], 0
        for (Co_, _, _, _, _), bs_ in zip(shapes, batch_stats):
            zoff.append(ztot)
            ztot += Co_ if bs_ else 0
        if zbuf is None or zbuf.numel() != ztot:      # (the fp16 path's forward prologue has cleared one of the right size)
            zbuf = torch.zeros(ztot, **f32) if ztot else None

        # last block: pool -> ReLU mask -> BatchNorm-backward sums
        Co, Ci, k, Tin, Tout = shapes[-1]
        a, bb, mean, invstd = affine[-1]
        y = inputs[-1]
        g = torch.empty_like(y)
        nparts = int(L.ign_bn_relu_pool_bwd_parts(B, Tout))
        part = torch.empty(nparts, 2, Co, **f32)
        _lib.check(L.ign_bn_relu_pool_bwd(_ptr(y), _ptr(gpool), _ptr(a), _ptr(bb), _ptr(mean), _ptr(invstd), _ptr(g), _ptr(part),
                                          B, Tout, Co, _stream()), "ign_bn_relu_pool_bwd")
        # weight-gradient partials of every block are reduced by ONE launch at the end (split-bf16 kernels with k in 2,3,5,8)
        defer = x6 and all(sh[2] in (2, 3, 5, 8) for sh in shapes) and nl <= 8
        deferred = []
        for l in range(nl - 1, -1, -1):
            Co, Ci, k, Tin, Tout = shapes[l]
            a, bb, mean, invstd = affine[l]
            y = inputs[l + 1]
            dbeta, dgamma = torch.empty(Co, **f32), torch.empty(Co, **f32)
            _lib.check(L.ign_bn_finalize_bwd(_ptr(part), nparts, Co, _ptr(dbeta), _ptr(dgamma), _stream()), "ign_bn_finalize_bwd")
            pad = (k - 1) if l > 0 else 0
            dyp = torch.empty(B, Tout + 2 * pad, Co, **f32)
            training = 1 if batch_stats[l] else 0
            # (fp16 path: the kernel also takes max |dL/dy| as it writes the tensor -- the bound its two consumers scale by)
            _lib.check(L.ign_bn_bwd_apply_amax(_ptr(g), _ptr(y), _ptr(a), _ptr(mean), _ptr(invstd), _ptr(dbeta), _ptr(dgamma), _ptr(dyp),
                                               fp(slots, 4 * l + 2) if h3 else None, B, Tout, Co, pad, training, _stream()),
                       "ign_bn_bwd_apply")
            if DEBUG is not None:
                DEBUG[f"g{l}"], DEBUG[f"dyp{l}"], DEBUG[f"dbeta{l}"], DEBUG[f"dgamma{l}"] = g.clone(), dyp.clone(), dbeta.clone(), dgamma.clone()
            del g
            # weight gradient; the operand below is relu(bn(y_{l-1})) recomputed on the fly (raw x for the first block)
            pa, pb = (affine[l - 1][0], affine[l - 1][1]) if l > 0 else (None, None)
            wx6 = x6 and k in (2, 3, 5, 8)
            ws_bytes = (L.ign_clconv_wgrad_x6_workspace_bytes if wx6 else L.ign_clconv_wgrad_workspace_bytes)(B, Tin, Ci, Co, k)
            ws = torch.empty(int(ws_bytes) // 4, **f32)
            dw = torch.empty(Co, Ci, k, **f32)
            wgrad = (L.ign_clconv_wgrad_bf16 if math == "bf16" else L.ign_clconv_wgrad_x6) if wx6 else L.ign_clconv_wgrad
            if h3 and wx6:
                _lib.check(L.ign_clconv_wgrad_h3(_ptr(dyp), pad, _ptr(inputs[l]), _ptr(pa), _ptr(pb), None if defer else _ptr(dw), _ptr(ws),
                                                 fp(slots, 4 * l + 2), fp(slots, 4 * l + 1) if l else _ptr(xbound), B, Tin, Ci, Co, k, _stream()),
                           "ign_clconv_wgrad_h3")
            else:
                _lib.check(wgrad(_ptr(dyp), pad, _ptr(inputs[l]), _ptr(pa), _ptr(pb), None if defer else _ptr(dw), _ptr(ws), B, Tin,
                                 Ci, Co, k, _stream()), "ign_clconv_wgrad")
            if defer:
                deferred.append((ws, dw, int(L.ign_clconv_wgrad_x6_nsplit(B, Tin, Ci, Co, k)), Co, Ci, k))
            del ws
            grads[4 * l + 0] = dw
            # bias: zero with batch statistics (see above); with running statistics it is the column sum of dL/dy
            grads[4 * l + 1] = zbuf[zoff[l]:zoff[l] + Co] if training else dyp.sum(dim=(0, 1))
            grads[4 * l + 2] = dgamma
            grads[4 * l + 3] = dbeta
            if l > 0:
                pa_, pb_, pm_, pi_ = affine[l - 1]
                g = torch.empty(B, Tin, Ci, **f32)
                nparts = int(L.ign_clconv_x6_mtiles(B, Tin) if x6 else L.ign_clconv_mtiles(B * Tin))
                part = torch.empty(nparts, 2, Ci, **f32)
                dgrad = (L.ign_clconv_dgrad_bf16 if math == "bf16" else L.ign_clconv_dgrad_x6) if x6 else L.ign_clconv_dgrad
                if h3:
                    _lib.check(L.ign_clconv_dgrad_h3(_ptr(dyp), _ptr(wds[l]), _ptr(inputs[l]), _ptr(pa_), _ptr(pb_), _ptr(pm_), _ptr(pi_),
                                                     _ptr(g), _ptr(part), fp(slots, 4 * l + 2), fp(slots, 4 * l), B, Tin, Ci, Co, k,
                                                     _stream()), "ign_clconv_dgrad_h3")
                else:
                    _lib.check(dgrad(_ptr(dyp), _ptr(wds[l]), _ptr(inputs[l]), _ptr(pa_), _ptr(pb_), _ptr(pm_), _ptr(pi_),
                                     _ptr(g), _ptr(part), B, Tin, Ci, Co, k, _stream()), "ign_clconv_dgrad")
            del dyp
        if deferred:
            n = len(deferred)
            vpa, ia = ctypes.c_void_p * n, ctypes.c_int * n
            _lib.check(L.ign_clconv_wgrad_reduce_multi(n, vpa(*[d[0].data_ptr() for d in deferred]), vpa(*[d[1].data_ptr() for d in deferred]),
                                                       ia(*[d[2] for d in deferred]), ia(*[d[3] for d in deferred]),
                                                       ia(*[d[4] for d in deferred]), ia(*[d[5] for d in deferred]), _stream()),
                       "ign_clconv_wgrad_reduce_multi")
        ctx.saved = None
        return (None, None, *grads, *head_grads)


def fcn_body(x, blocks, math=None, head=None):
    """x (B,T,C) float32 on the GPU; blocks = [(conv1d, batchnorm1d), ...] -> pooled (B, C_last).
    `math`: arithmetic of the GEMMs (see CONV_MATH); default = "bf16" inside an autocast region, else CONV_MATH.
    `head`: an nn.Linear over the pooled channels (<= 16 outputs) -> its logits (B, N) are returned instead, computed by the
    pooling launch."""
    if math is None:
        math = "bf16" if torch.is_autocast_enabled() else CONV_MATH
    states = [BnState(bn) for _, bn in blocks]            # (the step counters are bumped by the node's prologue launch)
    params = []
    for conv, bn in blocks:
        if conv.bias is None or bn.weight is None or bn.bias is None:
            raise _lib.IgnError("fcn_body: Conv1d needs a bias and BatchNorm1d needs affine parameters")
        if conv.stride != (1,) or conv.padding != (0,) or conv.dilation != (1,) or conv.groups != 1:
            raise _lib.IgnError("fcn_body: only stride-1, unpadded, undilated, ungrouped Conv1d is implemented")
        params += [conv.weight, conv.bias, bn.weight, bn.bias]
    if head is not None:
        if head.weight.shape[0] > 16:
            raise _lib.IgnError("fcn_body: the fused head covers up to 16 outputs")
        # (a head without bias passes a None placeholder: autograd.Function accepts non-tensor inputs)
        return FcnBodyFn.apply(x, (states, math, True), *params, head.weight, head.bias)
    return FcnBodyFn.apply(x, (states, math, False), *params)

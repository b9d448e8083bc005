"""ctypes loader for libign_hip.so -- the only door between the Python host side and the HIP kernels.

There is no CPU fallback: if the shared library is missing or a tensor is not on the GPU the call raises.
Build with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C speech-imagery-eeg_amd/csrc``.
"""
import ctypes
import os
import threading

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
_LOCK = threading.Lock()
_LIB = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i32 = ctypes.POINTER(ctypes.c_int32)
vp, ci, cf, sz, ll, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_longlong, ctypes.c_double

# name -> (restype, argtypes); must list every symbol include/ign_abi.h declares (tests/test_abi_symbols.py)
SIGNATURES = {
    "ign_abi_version": (ci, []),
    "ign_last_error": (ctypes.c_char_p, []),
    "ign_instnorm_fwd": (ci, [vp, vp, vp, ci, ci, ci, cf, vp]),
    "ign_transpose_btc_to_bct": (ci, [vp, vp, ci, ci, ci, vp]),
    "ign_instnorm_fwd_amax": (ci, [vp, vp, vp, ci, ci, ci, cf, vp, vp]),
    "ign_standardise_nct_to_btc": (ci, [vp, vp, vp, ci, ci, ci, cf, vp]),
    "ign_shapelet_fwd": (ci, [vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, ci, vp]),
    "ign_shapelet_fwd_bank": (ci, [vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, cf, ci, vp]),
    "ign_shapelet_bwd_workspace_bytes": (sz, [ci, ci, ci, ci, ci, ci, ci]),
    "ign_layernorm_parts": (ll, [ll, ci]),
    "ign_layernorm_fwd": (ci, [vp, vp, vp, vp, vp, vp, ll, ci, cf, vp]),
    "ign_bn1_data_stats_workspace_bytes": (sz, [ci, ci, ci]),
    "ign_bn1_data_stats": (ci, [vp, ci, ci, ci, ci, vp, vp, vp, vp]),
    "ign_autocorr_sum_fwd": (ci, [vp, vp, vp, ci, ci, ci, ctypes.POINTER(ctypes.c_int), vp]),
    "ign_bn1_fold_fwd": (ci, [vp, vp, vp, vp, vp, vp, cd, cf, cf, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "ign_bn1_fold_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, cd, vp, vp, vp, vp, ci, ci, ci, vp]),
    "ign_layernorm_res_fwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ll, ci, cf, vp]),
    "ign_layernorm_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ll, ci, vp]),
    "ign_layernorm_bwd_amax": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ll, ci, vp]),
    "ign_autocorr_parts": (ll, [ci]),
    "ign_autocorr_fwd": (ci, [vp, vp, ci, ci, ci, vp]),
    "ign_edge_lagprod_parts": (ll, [ci]),
    "ign_edge_lagprod_fwd": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "ign_attn_fwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ll, ll, cf, vp]),
    "ign_attn_fwd_x6": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ll, ll, cf, vp]),
    "ign_attn_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ll, ll, cf, vp]),
    "ign_attn_bwd_x6": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ll, ll, cf, vp]),
    "ign_attn_bwd_x6_strided": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ll, ll, cf, vp, ll, ll,
                                     ci]),
    "ign_attn_fwd_bf16": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ll, ll, cf, vp]),
    "ign_attn_bwd_bf16": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ll, ll, cf, vp]),
    "ign_attn_fwd_h3": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ll, ll, cf, vp, vp, vp, vp]),
    "ign_attn_bwd_h3": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ll, ll, ll, ll, ll, ll, cf, vp, ll, ll,
                             vp, vp, vp, vp, vp]),
    "ign_head_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ll, vp]),
    "ign_head_bwd": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, ll, vp]),
    "ign_gate_fwd": (ci, [vp, vp, vp, vp, ci, ci, cf, ci, vp]),
    "ign_gate_bwd": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, cf, ci, vp]),
    "ign_loss_fwd_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, vp]),
    "ign_diversity_fwd_bwd": (ci, [vp, vp, vp, ci, ci, ci, cf, vp]),
    "ign_adam_step": (ci, [vp, vp, vp, vp, ll, cf, cf, cf, cf, ci, vp]),
    "ign_gather_flat": (ci, [vp, vp, vp, ci, vp, vp]),
    "ign_adam_step_dev": (ci, [vp, vp, vp, vp, ll, cf, cf, cf, cf, vp, vp, vp]),
    "ign_conv1_sumsq_workspace_bytes": (sz, [ci, ci, ci]),
    "ign_conv1_sumsq_fwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_conv1_sumsq_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_dwconv1d_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]),
    "ign_dwconv1d_bwd_weight_workspace_bytes": (sz, [ci, ci, ci]),
    "ign_dwconv1d_bwd_weight": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_chan_contract_fwd": (ci, [vp, vp, vp, ci, ci, ci, ci, vp]),
    "ign_chan_contract_bwd_weight_workspace_bytes": (sz, [ci, ci, ci, ci]),
    "ign_chan_contract_bwd_weight": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "ign_chan_stats_workspace_bytes": (sz, [ci, ci]),
    "ign_chan_stats": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "ign_affine_elu_pool_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "ign_bn_elu_pool_bwd_sums": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "ign_bn_elu_pool_bwd_apply": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "ign_bn_fold_fwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ll, cf, cf, vp]),
    "ign_bn_fold_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ll, cf, vp]),
    "ign_clconv_mtiles": (ll, [ll]),
    "ign_clconv_pack_weights": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "ign_clconv_fwd": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_clconv_dgrad": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_clconv_kpad": (ci, [ci]),
    "ign_clconv_x6_mtiles": (ll, [ci, ci]),
    "ign_clconv_x3_elems": (ll, [ci, ci, ci]),
    "ign_clconv_pack_weights_x3": (ci, [vp, vp, vp, ci, ci, ci, vp]),
    "ign_clconv_fwd_x6": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_clconv_dgrad_x6": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_clconv_wgrad_x6_workspace_bytes": (sz, [ci, ci, ci, ci, ci]),
    "ign_clconv_wgrad_x6": (ci, [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_clconv_fwd_bf16": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_clconv_dgrad_bf16": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_clconv_wgrad_bf16": (ci, [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_linear_wgrad_x6": (ci, [vp, vp, vp, vp, vp, ll, ci, ci, vp]),
    "ign_linear_wgrad_bf16": (ci, [vp, vp, vp, vp, vp, ll, ci, ci, vp]),
    "ign_clconv_wgrad_workspace_bytes": (sz, [ci, ci, ci, ci, ci]),
    "ign_clconv_wgrad": (ci, [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_bn_finalize_fwd": (ci, [vp, ci, ll, ci, vp, vp, cf, cf, vp, vp, vp, vp, vp, vp, vp]),
    "ign_bn_affine_eval": (ci, [vp, vp, vp, vp, cf, ci, vp, vp, vp, vp, vp]),
    "ign_bn_relu_pool_fwd": (ci, [vp, vp, vp, vp, ci, ci, ci, vp]),
    "ign_bn_relu_pool_head_fwd": (ci, [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]),
    "ign_bn_relu_pool_bwd_parts": (ll, [ci, ci]),
    "ign_bn_relu_pool_bwd": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    "ign_bn_finalize_bwd": (ci, [vp, ci, ci, vp, vp, vp]),
    "ign_bn_bwd_apply": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_timing_enable": (ci, [ci]),
    "ign_timing_read": (ci, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]),
    "ign_clconv_wgrad_x6_nsplit": (ci, [ci, ci, ci, ci, ci]),
    "ign_clconv_wgrad_reduce_multi": (ci, [ci, vp, vp, vp, vp, vp, vp, vp]),
    "ign_clconv_pack_weights_x3_multi": (ci, [ci, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ign_absmax": (ci, [vp, ll, vp, vp]),
    "ign_absmax_multi": (ci, [ci, vp, vp, vp, vp]),
    "ign_fcn_scan": (ci, [ci, vp, vp, vp, vp, vp, vp, vp, vp, ll, vp]),
    "ign_clconv_pack_weights_h2_multi": (ci, [ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ign_clconv_fwd_h3": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_clconv_fwd_h3_amax": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_linear_dgrad_gelu_h3": (ci, [vp, vp, vp, vp, vp, vp, vp, ll, ci, ci, vp]),
    "ign_clconv_dgrad_h3": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_clconv_wgrad_h3": (ci, [vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_linear_wgrad_h3": (ci, [vp, vp, vp, vp, vp, vp, vp, ll, ci, ci, vp]),
    "ign_bn_bwd_apply_amax": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]),
    "ign_head_bwd_acc": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ll, vp]),
    "ign_loss_fwd_bwd_reg": (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, cf, vp]),
    "ign_sbm_reg_workspace_bytes": (sz, [ci, ci, ll]),
    "ign_sbm_reg_fwd_bwd": (ci, [vp, vp, ll, cf, ci, vp, vp, vp, vp, ci, cf, cf, vp, vp, vp]),
    "ign_shapelet_bwd_bank_workspace_bytes": (sz, [ci, ci, ci, ci, vp, vp, vp, ci]),
    "ign_shapelet_bwd_bank": (ci, [vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, cf, ci,
                                   vp]),
    "ign_shapelet_bwd": (ci, [vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cf, ci, vp]),
}


class IgnError(RuntimeError):
    pass


_RAW_STREAM = None

# Bumped whenever parameters are rewritten behind autograd's back (ign_adam_step writes through raw pointers, which does not move
# the tensors' version counters): anything derived from parameter VALUES and kept across calls is validated against it.
PARAM_GENERATION = [0]


def stream():
    """hipStream_t of torch's current stream on the current device, as the `stream` argument of the entry points.
    `torch.cuda.current_stream()` builds a Python Stream object each time (~9 us: 19 calls = 0.18 ms of a 0.9 ms step at
    run_uea.sh's batch 32); the raw getter behind it returns the handle in well under a microsecond."""
    global _RAW_STREAM
    import torch
    if _RAW_STREAM is None:
        _RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _RAW_STREAM:
        return ctypes.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def lib_path():
    return os.path.join(_CSRC, "libign_hip.so")


def lib():
    """Load (once) and return the ctypes handle; raises IgnError if the library was not built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOCK:
        if _LIB is None:
            path = lib_path()
            if not os.path.exists(path):
                raise IgnError(f"{path} not found: the HIP extension is not built "
                               f"(run `make -C {_CSRC} -j8` or __graft_entry__.build()); there is no CPU fallback")
            h = ctypes.CDLL(path)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(h, name)
                fn.restype, fn.argtypes = res, args
            if h.ign_abi_version() != 1:
                raise IgnError(f"{path}: ABI version {h.ign_abi_version()} != 1")
            _LIB = h
    return _LIB


def require():
    return lib()


def check(rc, what):
    if rc != 0:
        msg = lib().ign_last_error().decode("utf-8", "replace")
        raise IgnError(f"{what} failed (rc={rc}): {msg}")


def timing_enable(on=True):
    lib().ign_timing_enable(1 if on else 0)


def timing_read(label):
    """-> (total device ms, launches) of the kernels recorded under `label` since timing_enable(True)."""
    ms, n = ctypes.c_double(0.0), ctypes.c_longlong(0)
    check(lib().ign_timing_read(label.encode(), ctypes.byref(ms), ctypes.byref(n)), "ign_timing_read")
    return ms.value, n.value

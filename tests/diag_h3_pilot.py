"""Pilot of the two-plane fp16 convolution GEMM (ign_clconv_fwd_h3): speed against the six-product bf16 kernel and error of
both against float64, on the three FCN layers of the benchmark shape.   python tests/diag_h3_pilot.py [B]"""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    import speech_imagery_eeg_amd  # noqa: F401
    from ign_hip import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    ptr = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(0)
    for (Tin, Ci, Co, k, pro, xs) in [(1000, 122, 128, 8, False, 1.0), (993, 128, 256, 5, True, 3.0), (989, 256, 128, 3, True, 1e-3)]:
        x = (torch.randn(B, Tin, Ci, generator=g) * xs).to(dev)
        w = (torch.randn(Co, Ci, k, generator=g) / (Ci * k) ** 0.5).to(dev)
        bias = torch.randn(Co, generator=g).to(dev) * 0.1
        pa = (torch.rand(Ci, generator=g) + 0.5).to(dev) if pro else None
        pb = (torch.randn(Ci, generator=g) * 0.3 * xs).to(dev) if pro else None
        Tout = Tin - k + 1
        zin = torch.relu(pa * x + pb) if pro else x
        # float64 reference on 2 samples
        ref = F.conv1d(zin[:2].double().cpu().permute(0, 2, 1), w.double().cpu(), bias.double().cpu()).permute(0, 2, 1)
        res = {}
        for mode in ("x6", "h3"):
            n3 = int(L.ign_clconv_x3_elems(Co, Ci, k))
            wt = torch.zeros(n3, device=dev, dtype=torch.bfloat16)
            y = torch.empty(B, Tout, Co, device=dev)
            slots = torch.zeros(4, device=dev)
            slots[0] = w.abs().max()
            slots[1] = zin.abs().max() * 1.7            # any upper bound
            vp1, i1 = ctypes.c_void_p * 1, ctypes.c_int * 1
            if mode == "x6":
                _lib.check(L.ign_clconv_pack_weights_x3(ptr(w), ptr(wt), None, Co, Ci, k, st), "pack")
                run = lambda: _lib.check(L.ign_clconv_fwd_x6(ptr(x), ptr(wt), ptr(bias), ptr(pa), ptr(pb), ptr(y), None, B, Tin, Ci, Co,
                                                             k, st), "fwd_x6")
            else:
                _lib.check(L.ign_clconv_pack_weights_h2_multi(1, vp1(w.data_ptr()), vp1(wt.data_ptr()), None, i1(Co), i1(Ci), i1(k), None,
                                                              vp1(slots.data_ptr()), st), "pack_h2")
                b_in = ctypes.c_void_p(slots.data_ptr() + 4)
                run = lambda: _lib.check(L.ign_clconv_fwd_h3(ptr(x), ptr(wt), ptr(bias), ptr(pa), ptr(pb), ptr(y), None, b_in,
                                                             ptr(slots), B, Tin, Ci, Co, k, st), "fwd_h3")
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            err = float((y[:2].double().cpu() - ref).abs().max() / ref.abs().max())
            res[mode] = (ms, err)
        fl = 2.0 * B * Tout * Co * Ci * k
        print(f"Tin {Tin} Ci {Ci} Co {Co} k {k} pro {pro} xscale {xs:g}:  " +
              "  ".join(f"{m}: {v[0]:.3f} ms = {fl / v[0] / 1e9:.0f} TFLOP/s-eq, err vs f64 {v[1]:.2e}" for m, v in res.items()) +
              f"   speed-up {res['x6'][0] / res['h3'][0]:.2f}x")


if __name__ == "__main__":
    main()

"""Interpretability-gated mixture of the SBM expert and a deep expert (IGN/model/InterpGN.py:13-66)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ign_hip import ops
from models.FullyConvNet import FullyConvNetwork
from models.Shapelet import ShapeBottleneckModel
from utils.shapelet_util import ModelInfo


def _lazy(name):
    def ctor(configs):
        import importlib
        return importlib.import_module(f"models.{name}").Model(configs)
    return ctor


# registry keys of IGN/model/InterpGN.py:13-19
dnn_dict = {
    'PatchTST': _lazy('PatchTST'),
    'FCN': FullyConvNetwork,
    'TimesNet': _lazy('TimesNet'),
    'Transformer': _lazy('Transformer'),
    'ResNet': _lazy('ResNet'),
}


_SIDE_STREAMS = {}          # device -> HIP stream the deep expert runs on (see InterpGN._experts)


def gini_gate(sbm_out, deep_out, gating_value=None):
    """eta = (N * sum softmax(sbm)^2 - 1) / (N - 1); out = eta*sbm + (1-eta)*dnn   (InterpGN.py:44-52).
    At test time ``gating_value`` snaps eta to 1 where it exceeds the threshold."""
    q = F.softmax(sbm_out, dim=-1)
    n = sbm_out.shape[-1]
    eta = (n * q.pow(2).sum(-1, keepdim=True) - 1) / (n - 1)
    if gating_value is not None:
        hard = (eta > gating_value).float()
        eta = hard + eta * (1 - hard)
    return eta * sbm_out + (1.0 - eta) * deep_out, eta


class InterpGN(nn.Module):
    def __init__(self, configs, num_shapelet=[5, 5, 5, 5], shapelet_len=[0.1, 0.2, 0.3, 0.5]):
        super().__init__()
        self.configs = configs
        self.sbm = ShapeBottleneckModel(configs=configs, num_shapelet=num_shapelet, shapelet_len=shapelet_len)
        self.deep_model = dnn_dict[configs.dnn_type](configs)

    # The two experts are independent until the gate, so they are issued on two HIP streams (autograd replays each backward
    # node on its forward stream).  Measured (csrc/tools/coissue_microbench.hip): a SIMD does NOT execute another
    # wave's VALU instructions under an MFMA stream -- co-resident MFMA-bound and VALU-bound kernels finish in the SUM of their
    # times -- so what the second stream hides is only the memory- / latency-bound glue and the tails of the grids
    # (~1 ms of a 13.4 ms step), not the convolution GEMMs themselves.
    expert_streams = os.environ.get("IGN_EXPERT_STREAMS", "1") != "0"
    # Below this many input elements (B*T*C) a step is launch-bound, not kernel-bound (run_uea.sh trains at B = 32 on series of a
    # few thousand samples): the experts then run on ONE stream -- nothing to overlap, no fork / join events per step, and a
    # captured hipGraph of the step (ign_hip/graph.py) is a plain chain instead of two branches that replay serialised.
    two_stream_min_elems = 1 << 21

    def _experts(self, x, x_mark_enc, x_dec, x_mark_dec, mask):
        if not (self.expert_streams and x.is_cuda and x.numel() >= self.two_stream_min_elems):
            # one stream, SBM first: its instance-norm pass hands the FCN expert the magnitude bound of the raw batch
            self.sbm.input_bound = isinstance(self.deep_model, FullyConvNetwork) and x.is_cuda and x.dtype == torch.float32 \
                and not torch.is_autocast_enabled()
            sbm_out, info = self.sbm(x)
            return sbm_out, info, self.deep_model(x, x_mark_enc, x_dec, x_mark_dec, mask)
        self.sbm.input_bound = False                # two streams: the FCN expert is enqueued first and scans x itself
        main = torch.cuda.current_stream(x.device)
        side = _SIDE_STREAMS.get(x.device)
        if side is None:
            side = _SIDE_STREAMS[x.device] = torch.cuda.Stream(x.device, priority=-1)
        side.wait_stream(main)                      # x and the parameters are ready
        with torch.cuda.stream(side):
            deep_out = self.deep_model(x, x_mark_enc, x_dec, x_mark_dec, mask)
        x.record_stream(side)
        sbm_out, info = self.sbm(x)
        main.wait_stream(side)
        deep_out.record_stream(main)
        return sbm_out, info, deep_out

    def forward(self, x, x_mark_enc=None, x_dec=None, x_mark_dec=None, mask=None, gating_value=None):
        sbm_out, info, deep_out = self._experts(x, x_mark_enc, x_dec, x_mark_dec, mask)
        if torch.is_autocast_enabled():
            out, eta = gini_gate(sbm_out, deep_out, gating_value)
        else:
            out, eta = ops.gini_gate(sbm_out, deep_out, gating_value)
        # the SBM expert has just evaluated the regularisers on the same weights (IGN/model/Shapelet.py:206-210); the
        # reference recomputes them here (InterpGN.py:54-60) -- same value, same gradient, a dozen kernel launches less
        return out, ModelInfo(d=info.d, p=info.p, eta=eta, shapelet_preds=sbm_out, dnn_preds=deep_out,
                              preds=out, loss=info.loss, t=info.t)

    def loss(self):
        return self.sbm.loss()

    def step(self):
        self.sbm.step()

import os
import sys
from argparse import Namespace

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"fixture {name}.npz not generated")
    return dict(np.load(path, allow_pickle=False))


def make_cfg(**kw):
    """The argparse-namespace fields the reference's model ctors read."""
    base = dict(enc_in=6, seq_len=100, num_class=4, epsilon=1.0, distance_func='euclidean',
                memory_efficient=False, sbm_cls='linear', dropout=0.0, lambda_reg=0.1, lambda_div=0.1,
                dnn_type='FCN', task_name='classification', pred_len=0, label_len=0, output_attention=False,
                d_model=64, embed='timeF', freq='h', factor=1, n_heads=4, d_ff=128, activation='gelu',
                e_layers=2, c_out=4, dec_in=6, d_layers=1,
                eegcnn_layers=2, eegcnn_pooling='mean', eegcnn_dropout1=0.0, eegcnn_dropout2=0.0,
                eegcnn_n_heads=8, eegcnn_d_ff=256)
    base.update(kw)
    return Namespace(**base)


def sd_from(g, prefix="sd."):
    import torch
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(autouse=True)
def _reset_torch_matmul_precision():
    """Experiment.train() sets float32 matmul precision to 'medium' like the reference (exp:297); parity tests that
    run later in the same process must not inherit it."""
    import torch
    torch.set_float32_matmul_precision('highest')
    yield
    torch.set_float32_matmul_precision('highest')

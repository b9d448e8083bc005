"""Diagnostic (not a test): EEG-CNN baseline step with ops.layer_norm forced onto the HIP kernels at every row count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import speech_imagery_eeg_amd  # noqa
from ign_hip import ops
ops.LAYERNORM_MIN_ROWS = int(os.environ.get("LN_MIN_ROWS", "0"))
sys.argv = ["bench.py", "--config", "eegcnn", "--steps", "30", "--warmup", "5", "--cpu-sample", "0"]
import bench
bench.main()

"""MI355X-native hot path of the Interpretability-Gated-Network (drop-in for the reference's
``InterpretGatedNetwork/`` tree: same ``run.py`` flags, ``Experiment`` harness, model registry and
``data_provider`` contract; the shapelet block underneath runs hand-written HIP kernels for gfx950).

The reference is run from inside its own directory and imports ``models.*``, ``exp.*``, ``utils.*``,
``layers.*`` and ``data_provider.*`` as top-level packages (IGN/run.py:4-5, IGN/model/InterpGN.py:4-10).
This package keeps those names: importing it puts its directory on ``sys.path`` so the same absolute
imports resolve whether the entry point is ``python run.py`` from this directory (``run_uea.sh``) or
``import speech_imagery_eeg_amd`` from the repository root.
"""
import os
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
if PKG_DIR not in sys.path:
    sys.path.insert(0, PKG_DIR)

__version__ = "0.1.0"

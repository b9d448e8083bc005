"""Alias: the reference's directory is ``data_factory/`` while its code imports ``data_provider.*``."""
import sys
import data_provider as _m
sys.modules[__name__] = _m

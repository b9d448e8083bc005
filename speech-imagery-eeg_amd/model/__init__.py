"""Alias: the reference's directory is ``model/`` while its code imports ``models.*``."""
import sys
import models as _m
sys.modules[__name__] = _m

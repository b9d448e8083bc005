"""Python binding of libign_hip.so (ctypes over the C ABI in include/ign_abi.h) and the autograd ops on top."""
from ._lib import lib, require, lib_path, IgnError  # noqa: F401

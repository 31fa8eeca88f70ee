"""Loader for libdynamo_hip.so (the C-ABI HIP library built from ../csrc).

There is deliberately NO fallback: if the library is missing the product path raises.  The oracle
(oracle/) is test infrastructure and is never imported from here.
"""
import ctypes as C
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DYNAMO_HIP_LIB", os.path.join(_HERE, "libdynamo_hip.so"))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "csrc"))

_lib = None


class DynamoHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compiles every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", "8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise DynamoHipError("building libdynamo_hip.so failed")
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DynamoHipError(
                "libdynamo_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the loss path)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        abi.declare(lib)
        if lib.dd_abi_version() != abi.DD_ABI_VERSION:
            raise DynamoHipError("ABI mismatch between hipops/abi.py and libdynamo_hip.so")
        _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        msg = load().dd_error_string(code)
        raise DynamoHipError("%s failed: %s (%d)" % (what, msg.decode() if msg else "?", code))


def current_stream():
    """The raw hipStream_t PyTorch is currently enqueueing on (changes under graph capture / stream contexts)."""
    import torch
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))

"""The C ABI: libdynamo_hip.so builds for gfx950 on a GPU-less host, loads, and exports every symbol that
include/dynamo_hip.h declares; the ctypes mirror agrees with the header.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def so():
    from hipops import lib
    lib.build()
    return ctypes.CDLL(lib.LIB_PATH)


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "dynamo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dd_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported(so):
    from hipops import abi
    declared = header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(so, name), "libdynamo_hip.so does not export " + name
    assert sorted(abi.EXPORTED) == declared, (sorted(set(declared) ^ set(abi.EXPORTED)))


def test_abi_version_and_error_strings(so):
    from hipops import abi
    so.dd_abi_version.restype = ctypes.c_int
    assert so.dd_abi_version() == abi.DD_ABI_VERSION
    so.dd_error_string.restype = ctypes.c_char_p
    assert so.dd_error_string(0) == b"success"


def test_struct_layout_matches_header():
    """sizeof/offsetof computed by the C compiler vs the ctypes mirror."""
    import subprocess
    import tempfile
    from hipops import abi
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "dynamo_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(DDPhotoScale), sizeof(DDPhotoArgs), offsetof(DDPhotoArgs, scale),
         offsetof(DDPhotoScale, out_delta), offsetof(DDPhotoArgs, target), sizeof(DDAssembleArgs), offsetof(DDAssembleArgs, term_of));
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(DDRegSmooth), sizeof(DDRegScale), sizeof(DDRegArgs), offsetof(DDRegScale, w_sparsity),
         offsetof(DDRegScale, w_ground), offsetof(DDRegArgs, scale));
  printf("%zu %zu %zu %zu\n", sizeof(DDJpegHeader), offsetof(DDJpegHeader, qt), offsetof(DDJpegHeader, bits), offsetof(DDJpegHeader, vals));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "l.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "l")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = [int(x) for x in subprocess.check_output([exe]).split()]
    want = [ctypes.sizeof(abi.DDPhotoScale), ctypes.sizeof(abi.DDPhotoArgs), abi.DDPhotoArgs.scale.offset,
            abi.DDPhotoScale.out_delta.offset, abi.DDPhotoArgs.target.offset, ctypes.sizeof(abi.DDAssembleArgs),
            abi.DDAssembleArgs.term_of.offset,
            ctypes.sizeof(abi.DDRegSmooth), ctypes.sizeof(abi.DDRegScale), ctypes.sizeof(abi.DDRegArgs), abi.DDRegScale.w_sparsity.offset,
            abi.DDRegScale.w_ground.offset, abi.DDRegArgs.scale.offset,
            ctypes.sizeof(abi.DDJpegHeader), abi.DDJpegHeader.qt.offset, abi.DDJpegHeader.bits.offset, abi.DDJpegHeader.vals.offset]
    assert got == want, (got, want)


def test_loss_path_refuses_cpu_tensors():
    import torch
    import tools
    from hipops.lib import DynamoHipError
    with pytest.raises(DynamoHipError):
        tools.SSIM()(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))
    with pytest.raises(DynamoHipError):
        tools.disp_to_depth(torch.rand(1, 1, 8, 8), 0.1, 100.0)

"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/primia_hip.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os

import pytest

from primia_amd import _lib


def test_library_present_and_exports_header_symbols():
    assert os.path.exists(_lib.LIB_PATH), "run `python -m primia_amd.build` (or __graft_entry__.build())"
    protos = _lib.parse_header()
    assert len(protos) >= 30
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(lib, n)]
    assert not missing, f"header declares symbols the library does not export: {missing}"


def test_abi_version_and_host_queries():
    lib = _lib.lib()
    assert lib.primia_abi_version() >= 1
    d = _lib.ConvDesc.make(4, 56, 56, 64, 64, 3, 3, 1, 1)
    assert _lib.query("primia_conv_wfwd_elems", d) == 64 * 9 * 64
    assert _lib.query("primia_conv_wdgrad_elems", d) == 64 * 9 * 64
    stem = _lib.ConvDesc.make(4, 224, 224, 4, 64, 7, 7, 2, 3)
    assert (stem.Ho, stem.Wo) == (112, 112)
    assert _lib.query("primia_conv_wfwd_elems", stem) == 64 * 256
    bad = _lib.ConvDesc.make(4, 56, 56, 48, 64, 3, 3, 1, 1)  # channels not a multiple of 64
    assert _lib.query("primia_conv_wfwd_elems", bad) < 0


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from primia_amd.engine import ResNet18Engine

    with pytest.raises(_lib.PrimiaError):
        ResNet18Engine(batch_size=2)

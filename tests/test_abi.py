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


def test_option_table_epoch_and_probe_only_switches():
    """primia_set_option (host-side table, no GPU needed): the epoch counts CHANGES — restoring a value to what it already
    is does not invalidate engines that sized buffers under it — unknown names are refused, and the two
    timing-experiment switches that skip parts of a kernel (wrong results) are refused by the shipped library: only a probe
    build (python -m primia_amd.build --probe, loaded explicitly by tools/) honours them."""
    lib = _lib.lib()
    assert lib.primia_reset_options() == 0
    e0 = lib.primia_options_epoch()
    _lib.set_option("lh2", _lib.get_option("lh2"))
    assert lib.primia_options_epoch() == e0
    _lib.set_option("c64_blocks", 300)
    assert lib.primia_options_epoch() == e0 + 1 and _lib.get_option("c64_blocks") == 300
    assert lib.primia_reset_options() == 0 and lib.primia_options_epoch() == e0 + 2
    assert lib.primia_reset_options() == 0 and lib.primia_options_epoch() == e0 + 2
    with pytest.raises(_lib.PrimiaError):
        _lib.set_option("no_such_option", 1)
    for name in ("c64_dbg", "s2lh_dbg"):
        _lib.set_option(name, 0)
        with pytest.raises(_lib.PrimiaError, match="UNSUPPORTED"):
            _lib.set_option(name, 1)
        assert _lib.get_option(name) == 0

"""CPU-side checks of the drop-in boundary: the shared library builds, loads and exports every
symbol include/bsed.h declares.  No compute call is made here (no GPU in this tier)."""
import ctypes
import os

import pytest

import bsed_amd
from bsed_amd import _lib as L


def test_library_is_built_in_tree():
    assert os.path.exists(L.LIB_PATH), "run __graft_entry__.build() first"
    assert os.path.dirname(L.LIB_PATH).endswith("bird-sound-event-detecion_amd")


def test_every_header_symbol_is_exported():
    names = L.header_symbols()
    assert "bsed_mel_linear" in names and "bsed_last_error" in names
    lib = ctypes.CDLL(L.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/bsed.h but not exported: {missing}"


def test_error_convention_without_gpu():
    lib = L.lib()
    assert lib.bsed_abi_version() >= 1
    assert b"gfx950" in lib.bsed_build_info()
    # argument validation happens before any HIP call: NULL plan -> negative code + message
    rc = lib.bsed_mel_linear(None, None, 1, 32000, None, None, None, None, None)
    assert rc < 0 and b"null" in lib.bsed_last_error()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bsed_amd.features import MelFrontEnd
    with pytest.raises(L.BsedError):
        MelFrontEnd()

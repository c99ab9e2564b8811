"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and
exports every symbol include/sparsify_hip.h declares, with a ctypes signature for each."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "sparsify_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sphip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sparsify_amd import _lib
    assert _lib.LIB_PATH.exists(), "run `make -C sparsify_amd/csrc` first"
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(names)


def test_abi_version_and_error_reporting():
    from sparsify_amd import _lib
    lib = _lib.load()
    assert lib.sphip_abi_version() == _lib.ABI_VERSION
    # host-side validation happens before any launch: no GPU needed
    st = lib.sphip_topk(None, 0, None, None, 1, 8, 1, None, None, None)
    assert st == 1 and b"null pointer" in lib.sphip_last_error()
    st = lib.sphip_groupmax(1, 0, 4, 10, 3, 1, 1, None)
    assert st == 1 and b"not divisible" in lib.sphip_last_error()
    st = lib.sphip_encoder_preacts(16, 16, None, 16, None, 1, 4, 12, 8, 1, None)
    assert st == 2 and b"multiple of 8" in lib.sphip_last_error()
    with pytest.raises(_lib.HipLibraryError):
        _lib.check(st, "sphip_encoder_preacts")
    assert lib.sphip_segments_workspace_bytes(128, 16, 512) > 0
    assert lib.sphip_reduce_workspace_bytes(128, 64) > 0


def test_no_cpu_fallback():
    from sparsify_amd import SparseCoder, SparseCoderConfig, fused_encoder
    sae = SparseCoder(16, SparseCoderConfig(num_latents=64, k=4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sae(torch.randn(2, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fused_encoder(torch.randn(2, 16), torch.randn(64, 16), torch.randn(64), 4, "topk")


def test_product_never_imports_oracle():
    for p in (ROOT / "sparsify_amd").rglob("*.py"):
        src = p.read_text()
        assert "oracle" not in src.replace("# oracle", ""), f"{p} mentions the oracle"

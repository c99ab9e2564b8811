"""INTEGRATION.md's reference-side binding, executed as written: the ctypes stub a maintainer of the reference would
add (`sparsify/hip_backend.py` in the document) is cut out of the markdown, pointed at the in-tree library and its
`hip_decode` called the way the reference calls `decoder_impl(top_indices, top_acts, self.W_dec.mT)`
(sparse_coder.py:201), against torch's embedding_bag, which is what utils.eager_decode runs (utils.py:81-90)."""
import re
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_integration_snippet_runs_and_decodes(device):
    from sparsify_amd import _lib
    text = (ROOT / "INTEGRATION.md").read_text()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = next(b for b in blocks if "def hip_decode" in b)
    assert 'ctypes.CDLL("libsparsify_hip.so")' in stub
    stub = stub.replace('ctypes.CDLL("libsparsify_hip.so")', f'ctypes.CDLL("{_lib.LIB_PATH}")')
    ns: dict = {}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    g = torch.Generator().manual_seed(0)
    N, k, M, D = 300, 8, 1000, 96
    W_dec = torch.randn(M, D, generator=g).cuda()
    idx = torch.stack([torch.randperm(M, generator=g)[:k] for _ in range(N)]).cuda()
    acts = torch.rand(N, k, generator=g).cuda()
    got = ns["hip_decode"](idx, acts, W_dec.mT)
    torch.cuda.synchronize()
    want = torch.nn.functional.embedding_bag(idx, W_dec, per_sample_weights=acts, mode="sum")
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    # every argtypes list of the stub has the arity of the package's own binding of that entry point
    for name, (_, args) in _lib.SIGNATURES.items():
        fn = getattr(ns["_lib"], name, None)
        if fn is not None and getattr(fn, "argtypes", None):
            assert len(fn.argtypes) == len(args), name

"""Bounded, seeded slices of the randomised checks in benchmarks/*_fuzz.py, run inside the GPU suite
(the scripts' full runs are hours of cases; here each contributes the first CASES cases of its seeded
sequence, the same ones on every run). A mismatch raises AssertionError naming the case, which is all
one needs to replay it: `python benchmarks/<name>_fuzz.py` walks the same sequence.

  topk     every sphip_topk path against torch.topk: selections exact
  gemm     both GEMM tile kernels on small-integer operands: bit-exact, block maxima included
  sparse   decode gather + segmented weight gradients in all five kernel forms: 2e-5 of float64
  fused    fused encoder-gradient steps (SignSGD / schedule-free): bit for bit the separate path
  decoder  fused decoder steps: bit for bit the separate passes
  auxk     dense AuxK path against the row-gather decode: 2e-5
  long     training-size shapes with power-law usage through the weight gradients (the chunked long-segment
           path): 2e-5 of float64, fused steps bit for bit gradient + separate step, identical bits twice
"""
import importlib.util
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

BENCH = Path(__file__).resolve().parent.parent / "benchmarks"
CASES = {"topk": 200, "gemm": 150, "sparse": 80, "fused": 150, "decoder": 200, "auxk": 200, "long": 24}
ENV = ("SPHIP_GEMM", "SPHIP_GATHER", "SPHIP_WGRAD", "SPHIP_AUXK")


@pytest.mark.parametrize("name", list(CASES))
def test_fuzz_slice(device, name, monkeypatch):
    for var in ENV:                       # the scripts switch kernel forms through these: restored on teardown
        monkeypatch.delenv(var, raising=False)
    spec = importlib.util.spec_from_file_location(f"{name}_fuzz", BENCH / f"{name}_fuzz.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n = mod.run(budget=90.0, max_cases=CASES[name])
    # the time budget is only a safety net for a slow box: most of the slice must have run
    assert n >= CASES[name] // 2, f"{name}: only {n} of {CASES[name]} cases ran inside the time budget"

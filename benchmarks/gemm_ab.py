"""Interleaved A/B of sphip_encoder_preacts between library builds, in ONE process on one device
(boxes differ by more than the effects). Usage:
    python benchmarks/gemm_ab.py name=path[:ENV=VALUE,...] name=path ...
e.g. base=sparsify_amd/libsparsify_hip_base.so new=sparsify_amd/libsparsify_hip.so new_s1=sparsify_amd/libsparsify_hip.so:SPHIP_GEMM_SCHED=1
Prints the median / minimum of the rounds per arm and whether the arms' outputs are bit-identical."""
import ctypes
import os
import sys

import torch

N, D, M = 16384, 4096, 131072
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, D, device=dev, generator=g).to(torch.bfloat16)
W = (torch.randn(M, D, device=dev, generator=g) / 64).to(torch.bfloat16)
b = (torch.randn(M, device=dev, generator=g) * 0.1).to(torch.bfloat16)
pre = torch.empty(N, M, dtype=torch.bfloat16, device=dev)
tmax = torch.empty(N // 64, M // 128, 64, dtype=torch.bfloat16, device=dev)

arms = []
for a in sys.argv[1:]:
    name, rest = a.split("=", 1)
    path, _, envs = rest.partition(":")
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.sphip_encoder_preacts.restype = ctypes.c_int
    lib.sphip_encoder_preacts.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] + [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.c_void_p]
    env = dict(e.split("=") for e in envs.split(",") if e)
    arms.append((name, lib, env))


def launch(lib, env):
    for k_, v in env.items():
        os.environ[k_] = v
    rc = lib.sphip_encoder_preacts(x.data_ptr(), W.data_ptr(), b.data_ptr(), pre.data_ptr(), tmax.data_ptr(), 1, N, D, M, 1,
                                   torch.cuda.current_stream().cuda_stream)
    for k_ in env:
        del os.environ[k_]
    assert rc == 0, rc


res = {n: [] for n, _, _ in arms}
sums = {}
for rnd in range(int(os.environ.get("ROUNDS", "5"))):
    for name, lib, env in arms:
        launch(lib, env)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            launch(lib, env)
        e1.record()
        torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 5)
        if rnd == 0:
            sums[name] = (pre.view(torch.int16).to(torch.int64).sum().item(), tmax.view(torch.int16).to(torch.int64).sum().item(),
                          pre[::97, ::89].clone())
for name, v in res.items():
    v = sorted(v)
    ms = v[len(v) // 2]
    print(f"{name:12s} median {ms:.3f} ms (min {v[0]:.3f})  {2.0 * N * D * M / ms / 1e9:.0f} TFLOP/s", flush=True)
first = arms[0][0]
for name in res:
    same = sums[name][:2] == sums[first][:2] and torch.equal(sums[name][2], sums[first][2])
    print(f"{name} vs {first}: outputs {'identical' if same else 'DIFFER'} (checksums {sums[name][:2]})")

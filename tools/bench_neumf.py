#!/usr/bin/env python
"""NeuMF training throughput on one MI355X (BASELINE configs[3]: ml-1m shape,
d=64, num_ng=4).  A step = daisy_neumf_step_grads + one dense-Adam pass over the flat parameters.

The MLP tower is GEMM work: the roofline is the fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md;
the parity mode computes in fp32).  FLOPs counted = the three GEMMs per layer (forward, d-input,
d-weight) x 2 rows per sample; gathers / scatter / Adam are not counted.

    python tools/bench_neumf.py > profiles/rNN_bench_neumf.txt
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from daisyrec_amd import ops

U, I, D, L = 6040, 3706, 64, 3            # ml-1m users / items; NeuMF d=64, 3 layers: 512->256->128->64
PEAK_TF = 157.3          # dense fp32 MFMA peak, TFLOP/s
PEAK_BF16_TF = 2500.0    # dense bf16 MFMA peak, TFLOP/s
dev = torch.device("cuda", 0)


def make_params():
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    dm = D << (L - 1)
    shapes = {"uG": (U, D), "iG": (I, D), "uM": (U, dm), "iM": (I, dm)}
    w = 2 * dm
    for l in range(1, L + 1):
        shapes[f"W{l}"], shapes[f"b{l}"] = (w // 2, w), (w // 2,)
        w //= 2
    shapes["Wp"], shapes["bp"] = (1, 2 * D), (1,)
    names = ops.neumf_param_names(L)
    total = sum(int(torch.tensor(shapes[k]).prod()) for k in names)
    flat = torch.randn(total, device=dev, generator=g) * 0.05
    gflat = torch.zeros_like(flat)
    p, gr, off = {}, {}, 0
    for k in names:
        n = int(torch.tensor(shapes[k]).prod())
        p[k], gr[k] = flat[off:off + n].view(shapes[k]), gflat[off:off + n].view(shapes[k])
        off += n
    return flat, gflat, p, gr


def gemm_flops_per_sample():
    dm = D << (L - 1)
    w, macs = 2 * dm, 0
    for _ in range(L):
        macs += w * (w // 2)
        w //= 2
    return 2 * macs * 3 * 2          # x2 flop/MAC, x3 GEMMs (fwd, dX, dW), x2 rows (pos, neg)


def run(B, steps, dropout=0.0, bf16=False):
    flat, gflat, p, gr = make_params()
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    u = torch.randint(0, U, (B,), device=dev, generator=g, dtype=torch.int32)
    i = torch.randint(0, I, (B,), device=dev, generator=g, dtype=torch.int32)
    j = torch.randint(0, I, (B,), device=dev, generator=g, dtype=torch.int32)
    ctx = ops.NeumfContext(2 * B, D, L, U, I)
    ctx.set_precision(int(bf16))
    def step(t):
        ctx.step_grads(p, gr, u, i, j, 0, 1e-3, 1e-3, dropout=dropout, seed=t)
        ops.adam_dense(flat, gflat, m, v, 1e-3, t)
    for t in range(1, 4):
        step(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(4, 4 + steps):
        step(t)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    fl = gemm_flops_per_sample() * B
    out = {"B": B, "steps": steps, "dropout": dropout, "precision": {0: "fp32", 1: "bf16 MFMA inputs, fp32 storage", 2: "bf16 storage"}[int(bf16)], "ms_per_step": ms, "samples_per_s": B / ms * 1e3,
           "mlp_gemm_TFLOPs": fl / ms / 1e9,
           # the peak a precision mode is priced against: fp32 MFMA for the parity mode, dense bf16 MFMA for the two bf16 modes
           ("frac_of_fp32_mfma_peak" if int(bf16) == 0 else "frac_of_bf16_mfma_peak"):
               fl / ms / 1e9 / (PEAK_TF if int(bf16) == 0 else PEAK_BF16_TF),
           "workspace_GB": ctx.nbytes / 1e9}
    print(json.dumps(out), flush=True)
    ctx.close()


def gemm_only():
    for (M, N, K) in [(131072, 256, 512), (131072, 128, 256), (131072, 64, 128), (1 << 20, 256, 512)]:
        A = torch.randn(M, K, device=dev)
        Bm = torch.randn(N, K, device=dev)
        ops.gemm_nt(A, Bm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm_nt(A, Bm)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tf = 2.0 * M * N * K / ms / 1e9
        print(json.dumps({"gemm_nt": [M, N, K], "ms": ms, "TFLOPs": tf, "frac_of_fp32_mfma_peak": tf / PEAK_TF}),
              flush=True)
        ops.gemm_nt(A, Bm, bf16=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            ops.gemm_nt(A, Bm, bf16=True)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tf = 2.0 * M * N * K / ms / 1e9
        print(json.dumps({"gemm_nt_bf16_inputs": [M, N, K], "ms": ms, "TFLOPs": tf,
                          "frac_of_bf16_mfma_peak": tf / 2500.0,
                          "operand_GBps": (M * K + N * K + M * N) * 4 / ms / 1e6}), flush=True)
        A16, B16 = A.to(torch.bfloat16), Bm.to(torch.bfloat16)
        ops.gemm_nt_bf16(A16, B16)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            ops.gemm_nt_bf16(A16, B16)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tf = 2.0 * M * N * K / ms / 1e9
        print(json.dumps({"gemm_nt_bf16_storage": [M, N, K], "ms": ms, "TFLOPs": tf,
                          "frac_of_bf16_mfma_peak": tf / 2500.0,
                          "operand_GBps": (M * K + N * K + M * N) * 2 / ms / 1e6}), flush=True)


def cpu_baseline(B=65536, steps=2):
    from oracle.torch_port import TorchNeuMF
    torch.manual_seed(0)
    mdl = TorchNeuMF(U, I, D, L)
    g = torch.Generator()
    g.manual_seed(1)
    bt = [(torch.randint(0, U, (B,), generator=g), torch.randint(0, I, (B,), generator=g),
           torch.randint(0, I, (B,), generator=g)) for _ in range(steps + 1)]
    mdl.step(*bt[0])
    t0 = time.perf_counter()
    for b in bt[1:]:
        mdl.step(*b)
    dt = time.perf_counter() - t0
    print(json.dumps({"cpu_baseline": {"value": steps * B / dt, "unit": "samples/s", "cores": torch.get_num_threads(),
                                       "kind": "port", "sample": f"{steps} Adam steps at B={B} (oracle/torch_port.py: "
                                       f"TorchNeuMF, stock nn.Embedding/Linear/autograd/optim.Adam), {dt:.1f}s"}}), flush=True)


if __name__ == "__main__":
    gemm_only()
    run(256, 200)
    run(4096, 100)
    run(65536, 20)
    run(262144, 8)
    run(65536, 20, dropout=0.5)
    run(65536, 20, bf16=1)
    run(262144, 8, bf16=1)
    run(65536, 20, bf16=2)
    run(262144, 8, bf16=2)
    if "--no-cpu" not in sys.argv:
        cpu_baseline()

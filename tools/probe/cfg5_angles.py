#!/usr/bin/env python3
"""proposed_algorithm_angles at the BASELINE configs[4] shape with block-Toeplitz pilots (block height 256), alone - for kernel tables."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
Imax = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, M, Gr, G2 = 64, 65536, 64, 4096
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, device=dev), torch.randn(*s, generator=g, device=dev))
cm = J.colmajor
A = cm(rnd(N, Gr) / np.sqrt(N))
B0 = rnd(256, M) / np.sqrt(G2)
Bt = torch.empty(G2, M, dtype=torch.complex64, device=dev)
for ld in range(16):
    Bt[256 * ld:256 * (ld + 1), ld:] = B0[:, :M - ld]
    if ld:
        Bt[256 * ld:256 * (ld + 1), :ld] = rnd(256, ld) / np.sqrt(G2)
Bt = cm(Bt); del B0
S0 = torch.zeros(batch, Gr, G2, dtype=torch.complex64, device=dev)
idx = torch.randint(0, Gr * G2, (batch, 40), generator=g, device=dev)
S0.view(batch, -1).scatter_(1, idx, rnd(batch, 40))
Om = (torch.rand(batch, N, M, generator=g, device=dev) < 0.125).float()
subY = torch.stack([Om[t] * (A @ S0[t] @ Bt + 0.05 * rnd(N, M)) for t in range(batch)])
indx = (torch.argsort(S0.transpose(1, 2).reshape(batch, -1).abs(), dim=1, descending=True, stable=True) + 1).to(torch.int32)
fro2 = (subY.abs() ** 2).sum(dim=(1, 2)).double().cpu().numpy()
tY = 1.0 / fro2; tS = np.full(batch, 1e-3); rho = np.full(batch, 0.2)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S, Y, ce = J.proposed_algorithm_angles(cm(subY), cm(Om), indx, A, Bt, Imax, tY, tS, rho, "approximate", None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("angles, block-Toeplitz pilots (block %d), batch %d, %d iterations: %.3f s = %.2f channel-estimates/s per GPU" % (
        J.default_context(0).last_dictionary_block(), batch, Imax, dt, batch / dt), flush=True)
import hashlib
print("sha1 S %s  Y %s" % (hashlib.sha1(S.cpu().numpy().tobytes()).hexdigest()[:16], hashlib.sha1(Y.cpu().numpy().tobytes()).hexdigest()[:16]))

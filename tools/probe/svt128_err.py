#!/usr/bin/env python3
"""svt / mc_svt at the configs[2] shape against the float64 oracle (16 trials): max relative error (JSTSP_MC_EIG_STOP study)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import jstsp19_amd as J
from oracle import solvers as O
N_, B = 128, 16
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1283)
rnd = lambda *s: torch.complex(torch.randn(*s, generator=g, device=dev), torch.randn(*s, generator=g, device=dev))
idx = torch.arange(N_, device=dev, dtype=torch.float64)
D = (torch.exp(-2j * np.pi * idx[:, None] * idx[None, :] / N_) / np.sqrt(N_)).to(torch.complex64)
Sp = torch.zeros(B, N_, N_, dtype=torch.complex64, device=dev)
Sp[:, ::17, ::13] = rnd(B, len(range(0, N_, 17)), len(range(0, N_, 13)))
H = D @ Sp @ D.conj().T
OH = H + 0.05 * rnd(B, N_, N_)
Om = (torch.rand(B, N_, N_, generator=g, device=dev) < 0.125).float()
cm = J.colmajor
h = lambda x, t, dt=np.complex128: x[t].cpu().numpy().astype(dt)
sv0 = torch.linalg.svdvals(OH[:4].to(torch.complex128))
tau = np.full(B, float(sv0[:, N_ // 3].mean()))
X = J.svt(cm(OH), tau)
e1 = max(np.max(np.abs(h(X, t) - O.svt(h(OH, t), tau[t]))) / np.max(np.abs(h(OH, t))) for t in range(B))
Xm = J.mc_svt(cm(Om * OH), cm(Om), 20, np.full(B, 0.05), np.full(B, 0.1))
e2 = 0.0
for t in range(B):
    oh, om = h(Om * OH, t), h(Om, t, np.float64)
    e2 = max(e2, np.max(np.abs(h(Xm, t) - O.mc_svt(oh, om, 20, 0.05, 0.1))) / np.max(np.abs(oh)))
Xa, ce = J.mc_admm(cm(H), cm(Om * OH), cm(Om), 20, np.full(B, 0.05), np.full(B, 0.1))
e3 = e4 = 0.0
for t in range(B):
    oh, om = h(Om * OH, t), h(Om, t, np.float64)
    Xo, ceo = O.mc_admm(h(H, t), oh, om, 20, 0.05, 0.1)
    e3 = max(e3, np.max(np.abs(h(Xa, t) - Xo)) / np.max(np.abs(oh)))
    e4 = max(e4, np.max(np.abs(ce[t].cpu().numpy() - ceo) / np.abs(ceo)))
print("JSTSP_MC_EIG_STOP=%s: svt %.3e  mc_svt x20 %.3e  mc_admm x20 %.3e (ce %.3e)" % (os.environ.get("JSTSP_MC_EIG_STOP"), e1, e2, e3, e4))

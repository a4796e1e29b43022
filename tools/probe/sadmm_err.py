#!/usr/bin/env python3
"""sparse_admm at the configs[2] shape, 16 trials against the float64 oracle: max relative error of S and of convergence_error."""
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
cm = J.colmajor
S, ce = J.sparse_admm(cm(H), cm(OH), cm(D), cm(D), 100)
Dn = D.cpu().numpy().astype(np.complex128)
es, ec = [], []
for t in range(B):
    So, ceo = O.sparse_admm(H[t].cpu().numpy().astype(np.complex128), OH[t].cpu().numpy().astype(np.complex128), Dn, Dn, 100)
    es.append(np.max(np.abs(S[t].cpu().numpy() - So)) / np.max(np.abs(So)))
    ec.append(np.max(np.abs(ce[t].cpu().numpy() - ceo) / np.abs(ceo)))
print("JSTSP_M3_MINK=%s: S max %.3e mean %.3e; ce max %.3e mean %.3e" % (os.environ.get("JSTSP_M3_MINK"), max(es), np.mean(es), max(ec), np.mean(ec)))

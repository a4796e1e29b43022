#!/usr/bin/env python3
"""The plot_errorVSsnr.m sweep (reference-native parameters, :8-25) on the HIP path: proposed_algorithm,
proposed_algorithm_angles, LS, VAMP and MMV-OMP baselines (+ the commented TSSR recipe with --tssr); prints the mean
capped NMSE per SNR point, or with --rate the rate of plot_rateVSframelength.m:81."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jstsp19_amd.montecarlo import run_sweep
from jstsp19_amd.system_model import SweepParams

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=64)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--builder", default="hip", choices=["hip"], help="(the torch tensor-op builder moved to tests/torch_builder.py)")
ap.add_argument("--rate", action="store_true", help="rate metric of plot_rateVSframelength.m instead of the NMSE")
ap.add_argument("--tssr", action="store_true", help="add the TSSR recipe (mc_svt with rho = 0.1, then joint OMP)")
ap.add_argument("--vamp-large", action="store_true", help="VAMP column also where L*Gt > 128 (one order-L*Gt eigen-decomposition per trial)")
ap.add_argument("--config3", action="store_true", help="BASELINE configs[3]: Nt=Nr=64, Nrf=8, K=64, L=8, 10 SNR points")
ap.add_argument("--dist", action="store_true",
                help="one rank per GPU (start with python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 "
                     "tools/run_errorVSsnr.py --dist ...): (point, trial) pairs sharded, one RCCL all-reduce of the NMSE sums")
a = ap.parse_args()
dist = None
rank = 0
if a.dist:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    rank = dist.get_rank()
base = SweepParams(Nt=4, Nr=32, L=4, T=35, Mr=4)              # plot_errorVSsnr.m:8-23
snrs = list(range(-15, 16, 3))                                  # :24
if a.config3:
    base = SweepParams(Nt=64, Nr=64, L=8, T=64, Mr=8)           # the configs[1] shape (N=64, M=4096, Gr=64, G2=512)
    snrs = list(range(-15, 15, 3))                              # 10 points
t0 = time.perf_counter()
out = run_sweep(base, snrs, a.trials, Imax=100, batch=a.batch, baselines=True, numOfnz=100, builder=a.builder, vamp_max_order=8192 if a.vamp_large else 128,
                metric="rate" if a.rate else "nmse", tssr=(100, 0.1) if a.tssr else None, dist=dist)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
if rank != 0:
    sys.exit(0)
print("SNR(dB)  proposed  proposed+angles  LS        VAMP      MMV-OMP%s   (%s, %d trials/point, %s input builder, %.1f s)"
      % ("   TSSR      SVT" if a.tssr else "", "rate [bit/s/Hz]" if a.rate else "capped NMSE", a.trials, a.builder, dt))
for s, row in zip(snrs, out.tolist()):
    print("%6d   %.5f   %.5f          " % (s, row[0], row[1]) + "   ".join("%.5f" % v for v in row[2:]))

#!/usr/bin/env python3
"""What the host link of this box delivers: H2D / D2H from pinned and pageable memory, and the cost of pinning a
caller's pageable array in place (hipHostRegister) - the options a JSTSP_HOST call has for its 4.75 GiB of inputs."""
import ctypes as C, time, numpy as np, torch
hip = C.CDLL("libamdhip64.so")
n = 1 << 30
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pag = np.ones(n, dtype=np.uint8)
pt = torch.from_numpy(pag)
def t(fn, rep=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rep): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / rep
print("H2D pinned   : %.1f GB/s" % (n / t(lambda: dev.copy_(pin, non_blocking=True)) / 1e9))
print("D2H pinned   : %.1f GB/s" % (n / t(lambda: pin.copy_(dev, non_blocking=True)) / 1e9))
print("H2D pageable : %.1f GB/s" % (n / t(lambda: dev.copy_(pt)) / 1e9))
print("D2H pageable : %.1f GB/s" % (n / t(lambda: pt.copy_(dev)) / 1e9))
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
for rep in range(2):
    t0 = time.perf_counter(); rc = hip.hipHostRegister(pag.ctypes.data, n, 0); t1 = time.perf_counter()
    hip.hipMemcpyAsync(dev.data_ptr(), pag.ctypes.data, n, 1, None); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
    hip.hipHostUnregister(pag.ctypes.data); t3 = time.perf_counter()
    print("hipHostRegister rc=%d: %.1f GB/s; copy from registered: %.1f GB/s; unregister %.1f GB/s; all three: %.1f GB/s"
          % (rc, n / (t1 - t0) / 1e9, n / (t2 - t1) / 1e9, n / (t3 - t2) / 1e9, n / (t3 - t0) / 1e9))
# chunked pageable -> pinned bounce (memcpy by numpy, single thread) overlapped with H2D of the previous chunk
ch = 64 << 20
pn = pin.numpy()
def bounce():
    for o in range(0, n, ch):
        b = (o // ch) & 1
        pn[b * ch:(b + 1) * ch] = pag[o:o + ch]
        dev[o:o + ch].copy_(pin[b * ch:(b + 1) * ch], non_blocking=True)
print("pageable -> pinned bounce (1 thread, 64 MiB chunks) + H2D: %.1f GB/s" % (n / t(bounce, 2) / 1e9))

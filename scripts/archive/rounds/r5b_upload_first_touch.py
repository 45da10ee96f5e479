"""Round 5 (second session): why is the FIRST upload of a CSR from fresh host arrays 5-9x slower than the second (config 4: 178 ms against 19 ms for 1.07 GB)?
hipMemcpy host -> device of freshly written numpy arrays: first, second, third copy of the same array; a fresh array again; an array with MADV_HUGEPAGE set before it is written;
a registered (hipHostRegister) array; and a copy staged through two pinned 32 MB buffers filled by host threads."""
import ctypes as C, time, sys, os, mmap
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
N = 1 << 30
dev = torch.empty(N, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
def copy(a):
    t = time.perf_counter(); rc = hip.hipMemcpy(dev.data_ptr(), a.ctypes.data, a.nbytes, 1); assert rc == 0; return (time.perf_counter() - t) * 1e3
def fresh(huge=False):
    if huge:
        m = mmap.mmap(-1, N + (2 << 20)); m.madvise(mmap.MADV_HUGEPAGE)
        a = np.frombuffer(m, dtype=np.uint8, count=N, offset=0)
        a = a.view(); a.flags.writeable = True
    else: a = np.empty(N, dtype=np.uint8)
    a[:] = 7
    return a
for rep in range(2):
    a = fresh(); print("fresh array %d: copies of 1 GiB take %s ms" % (rep, ["%.1f" % copy(a) for _ in range(3)]), flush=True); del a
a = fresh(True); print("MADV_HUGEPAGE array: %s ms" % ["%.1f" % copy(a) for _ in range(3)], flush=True); del a
a = fresh(); t = time.perf_counter(); rc = hip.hipHostRegister(a.ctypes.data, a.nbytes, 0); reg = (time.perf_counter() - t) * 1e3
print("hipHostRegister rc %d took %.1f ms, then copies %s ms" % (rc, reg, ["%.1f" % copy(a) for _ in range(2)]), flush=True); hip.hipHostUnregister(a.ctypes.data); del a
# torch's pinned staging for comparison
a = fresh(); t = time.perf_counter(); x = torch.from_numpy(a).cuda(); torch.cuda.synchronize(); print("torch .cuda() of a fresh array: %.1f ms" % ((time.perf_counter() - t) * 1e3)); 
t = time.perf_counter(); x = torch.from_numpy(a).cuda(); torch.cuda.synchronize(); print("torch .cuda() again: %.1f ms" % ((time.perf_counter() - t) * 1e3))

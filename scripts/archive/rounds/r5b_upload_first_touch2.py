"""Is the slow first hipMemcpy per process or per address range?  Three arrays kept alive (three different address ranges), each copied three times; then hipHostRegister on a fourth, never-copied range."""
import ctypes as C, time
import numpy as np, torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
N = 1 << 30
dev = torch.empty(N, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
def copy(a):
    t = time.perf_counter(); rc = hip.hipMemcpy(dev.data_ptr(), a.ctypes.data, a.nbytes, 1); assert rc == 0; return (time.perf_counter() - t) * 1e3
keep = []
for rep in range(3):
    a = np.empty(N, dtype=np.uint8); a[:] = rep; keep.append(a)
    print("array %d at %#x: copies of 1 GiB take %s ms" % (rep, a.ctypes.data, ["%.1f" % copy(a) for _ in range(3)]), flush=True)
a = np.empty(N, dtype=np.uint8); a[:] = 9; keep.append(a)
t = time.perf_counter(); rc = hip.hipHostRegister(a.ctypes.data, a.nbytes, 0); reg = (time.perf_counter() - t) * 1e3
print("array 3 at %#x: hipHostRegister rc %d took %.1f ms, then copies %s ms" % (a.ctypes.data, rc, reg, ["%.1f" % copy(a) for _ in range(2)]), flush=True)
a = np.empty(N, dtype=np.uint8); a[:] = 5; keep.append(a)
def acopy(a):
    t = time.perf_counter(); rc = hip.hipMemcpyAsync(dev.data_ptr(), a.ctypes.data, a.nbytes, 1, None); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
print("array 4 at %#x: hipMemcpyAsync + sync %s ms" % (a.ctypes.data, ["%.1f" % acopy(a) for _ in range(3)]), flush=True)
a = np.empty(N // 8, dtype=np.uint8); a[:] = 5; keep.append(a)
print("array 5 (128 MiB) at %#x: hipMemcpy %s ms" % (a.ctypes.data, ["%.1f" % copy(a) for _ in range(3)]), flush=True)

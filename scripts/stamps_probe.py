"""Where does a wavefront of k_units spend its time?  Diagnostic build only (-DTILESPMV_STAMPS, a separate library under
gpurun_out/, never the product): lane 0 of every wavefront stamps the shader clock at kernel entry, after the task has arrived,
after the prologue / entry loads have arrived, after the entry phase, after the unit loop, after the stores were issued and after
they were acknowledged; s_memrealtime at entry gives the start skew between wavefronts.  Read the shares, not the total: the
waits forced at the stamps are not in the real kernel.     python scripts/stamps_probe.py <workload> [ENV=v ...]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "stamps"); os.makedirs(out, exist_ok=True)
src = os.path.join(ROOT, "tilespmv_amd", "csrc")
lib = os.path.join(out, "libtilespmv_f64.so")
hip = "/opt/rocm/bin/hipcc"
common = ["-O3", "-fPIC", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-DMAT_VAL_TYPE=double", "-DTILESPMV_STAMPS", "-Wno-unused-result"]
objs = []
for f in ("host_tile_create.cpp", "host_tilespmv_cpu.cpp", "host_mmio.cpp", "host_matrix_io.cpp", "host_multi.cpp"):
    o = os.path.join(out, f + ".o"); objs.append(o)
    subprocess.check_call([hip] + common + ["-ffp-contract=off", "-x", "c++", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-c", os.path.join(src, f), "-o", o])
for f in ("hip_plan.hip", "hip_plan_stream.hip", "hip_kernels.hip"):
    o = os.path.join(out, f + ".o"); objs.append(o)
    subprocess.check_call([hip] + common + ["--offload-arch=gfx950", "-munsafe-fp-atomics", "-c", os.path.join(src, f), "-o", o])
subprocess.check_call([hip, "-shared", "-fPIC", "--offload-arch=gfx950", "-pthread"] + objs + ["-ldl", "-o", lib])

from tilespmv_amd import _lib
_lib.lib_path = lambda dtype: lib            # this process only: the diagnostic library instead of the product's
import torch
from tilespmv_amd import api, generators as G
wl = sys.argv[1] if len(sys.argv) > 1 else "scircuit"
for kv in sys.argv[2:]:
    k, v = kv.split("="); os.environ[k] = v
sys.argv = sys.argv[:1]
import bench
m, n, rp, ci, _ = bench.build_matrix(wl)
rows = (m // 16) * 16; nnz = int(rp[rows])
vals, x = G.compat_values(len(ci)), G.compat_x(n)
tm = api.Tile_create(rows, n, nnz, rp, ci, vals)
plan = api.Plan(tm, rows, n, nnz)
xd = torch.from_numpy(x).cuda(); yd = torch.zeros(rows + 16, dtype=torch.float64, device="cuda")
for _ in range(20):
    plan.spmv(xd.data_ptr(), yd.data_ptr())
torch.cuda.synchronize()
ms = plan.time(xd.data_ptr(), yd.data_ptr(), warmup=5, reps=50)
plan.spmv(xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
info = plan.info()
nw = (info["num_tasks"] + 15) // 16 * 4
buf = np.zeros(nw * 8, dtype=np.uint64)
f = plan.lib.tilespmv_plan_stamps; f.restype = C.c_longlong; f.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
got = f(plan.h, buf.ctypes.data_as(C.c_void_p), buf.size)
assert got == buf.size, got
s = buf.reshape(nw, 8).astype(np.int64)
s = s[s[:, 0] > 0]
rt = (s[:, 7] - s[:, 7].min()) * 10e-3          # s_memrealtime: 100 MHz -> us
d = np.diff(s[:, :7], axis=1).astype(np.float64)  # cycles per segment
names = ["entry -> task arrived", "-> prologue + entry loads arrived", "-> entry phase done", "-> unit loop done", "-> stores issued", "-> stores acknowledged"]
life = (s[:, 6] - s[:, 0]).astype(np.float64)
print("%s: %.4f ms per SpMV in this (stamped) build, entry mode %d, %d tasks, %d wavefronts with work" % (wl, ms, info["entry_mode"], info["num_tasks"], len(s)))
print("wavefront start skew (s_memrealtime at entry): p50 %.2f us, p90 %.2f us, max %.2f us" % tuple(np.percentile(rt, [50, 90, 100])))
print("wavefront lifetime: median %.0f cycles, p90 %.0f, max %.0f" % tuple(np.percentile(life, [50, 90, 100])))
for i, nm in enumerate(names):
    print("  %-36s median %7.0f  p90 %7.0f  max %7.0f cycles  (%4.1f %% of the median lifetime)" % (nm, *np.percentile(d[:, i], [50, 90, 100]), 100 * np.median(d[:, i]) / np.median(life)))

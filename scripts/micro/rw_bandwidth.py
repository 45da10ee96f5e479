import torch
n = 1 << 28  # 2 GiB of f64
a = torch.zeros(n, dtype=torch.float64, device="cuda"); b = torch.ones(n, dtype=torch.float64, device="cuda")
def t(f, reps=10):
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
gb = n * 8 / 1e9
ms = t(lambda: a.fill_(1.0)); print("fill  (W only) %.3f ms  %.0f GB/s" % (ms, gb / ms * 1e3))
ms = t(lambda: a.copy_(b)); print("copy  (R+W)    %.3f ms  %.0f GB/s total" % (ms, 2 * gb / ms * 1e3))
ms = t(lambda: torch.sum(b)); print("sum   (R only) %.3f ms  %.0f GB/s" % (ms, gb / ms * 1e3))
ms = t(lambda: torch.add(a, b, out=a)); print("a+=b  (2R+W)   %.3f ms  %.0f GB/s total" % (ms, 3 * gb / ms * 1e3))

import torch, time
n = 1 << 29  # 4 GiB of f64
x = torch.empty(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.fill_(1.5)); print("fill  (write only) %.2f ms  %.0f GB/s" % (ms, n * 8 / ms / 1e6))
ms = t(lambda: x.zero_()); print("zero  (memset)     %.2f ms  %.0f GB/s" % (ms, n * 8 / ms / 1e6))
ms = t(lambda: x.sum()); print("sum   (read only)  %.2f ms  %.0f GB/s" % (ms, n * 8 / ms / 1e6))
ms = t(lambda: y.copy_(x)); print("copy  (r + w)      %.2f ms  %.0f GB/s" % (ms, 2 * n * 8 / ms / 1e6))
ms = t(lambda: torch.add(x, 1.0, out=y)); print("add   (r + w)      %.2f ms  %.0f GB/s" % (ms, 2 * n * 8 / ms / 1e6))
ms = t(lambda: torch.add(x, y, out=y)); print("add2  (2r + w)     %.2f ms  %.0f GB/s" % (ms, 3 * n * 8 / ms / 1e6))

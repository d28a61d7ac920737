"""How does a host -> device copy of the packed reads behave on this box: alone, and under a compute kernel on another
stream (two handles in flight feed the GPU from host pinned memory)?  torch only: plumbing."""
import time, torch
dev = torch.device("cuda", 0)
n = 138 * (1 << 20) // 4
h = torch.empty(n, dtype=torch.int32).pin_memory()
d = torch.empty(n, dtype=torch.int32, device=dev)
s_copy, s_comp = torch.cuda.Stream(), torch.cuda.Stream()
def t_copy(reps=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s_copy):
        for _ in range(reps):
            d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for _ in range(2): t_copy(2)
a = t_copy()
print("H2D 138 MiB alone: %.3f ms = %.1f GB/s" % (a * 1e3, n * 4 / a / 1e9))
x = torch.randn(8192, 8192, device=dev)
def busy(reps):
    with torch.cuda.stream(s_comp):
        y = x
        for _ in range(reps):
            y = torch.sin(y) * 1.0001
        return y
torch.cuda.synchronize(); t0 = time.perf_counter(); busy(20); torch.cuda.synchronize(); b = time.perf_counter() - t0
print("elementwise kernels alone (20 x 256 MiB): %.3f ms" % (b * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter()
busy(20)
with torch.cuda.stream(s_copy):
    for _ in range(3):
        d.copy_(h, non_blocking=True)
torch.cuda.synchronize(); c = time.perf_counter() - t0
print("both together: %.3f ms (sum %.3f, max %.3f)" % (c * 1e3, (b + 3 * a) * 1e3, max(b, 3 * a) * 1e3))
# D2H
t0 = time.perf_counter()
with torch.cuda.stream(s_copy):
    for _ in range(10):
        h.copy_(d, non_blocking=True)
torch.cuda.synchronize(); e = (time.perf_counter() - t0) / 10
print("D2H 138 MiB: %.3f ms = %.1f GB/s" % (e * 1e3, n * 4 / e / 1e9))

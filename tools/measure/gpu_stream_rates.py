"""plain streaming rates of this box (torch ops on 632 MB = the fp32 mask logits of a B = 64 frame batch): what can a write-only / copy kernel reach?"""
import torch, time
n = 64 * 3969 * 625
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty(n, dtype=torch.float32, device="cuda")
h = torch.empty(n * 2, dtype=torch.float16, device="cuda")
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
gb = n * 4 / 1e9
print("zero_  (write %.0f MB): %.0f GB/s" % (gb * 1e3, gb / t(lambda: a.zero_())))
print("fill_  (write): %.0f GB/s" % (gb / t(lambda: a.fill_(1.5))))
print("copy_  (read + write): %.0f GB/s total" % (2 * gb / t(lambda: b.copy_(a))))
print("sum    (read): %.0f GB/s" % (gb / t(lambda: a.sum())))
print("half->float copy (read 2 B, write 4 B per element): %.0f GB/s total" % (6 * n / 1e9 / t(lambda: b.copy_(h[:n]))))

"""Calibration: what plain streaming reaches on the box the profile ran on (torch copy / fill / sum of 1 GiB), next to which the step's
HBM-bound kernels (4.2-4.9 TB/s of counter traffic) are read."""
import torch
n = 1 << 29
a = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
b = torch.empty_like(a)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
gb = a.numel() * 2 / 1e12
print("copy 1 GiB -> 1 GiB : %.2f TB/s (read + write)" % (2 * gb / t(lambda: b.copy_(a))))
print("fill 1 GiB          : %.2f TB/s (write)" % (gb / t(lambda: b.zero_())))
print("sum  1 GiB (fp32 acc): %.2f TB/s (read)" % (gb / t(lambda: a.float().sum() if False else torch.sum(a, dtype=torch.float32))))
print("add  a + b -> b     : %.2f TB/s (2 reads + write)" % (3 * gb / t(lambda: torch.add(a, b, out=b))))

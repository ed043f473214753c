"""Times a plain fill and a plain copy of the gather output's size (Pavia training list, fp32):
the write-only and read+write floors the gather kernel can be compared with."""
import torch
n = 34220 * 927
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
def t(fn, reps=50):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
f = t(lambda: x.fill_(1.0)); c = t(lambda: y.copy_(x))
mb = n * 4 / 1e6
print(f"fill {mb:.0f} MB: {f:.1f} us = {mb/f:.2f} TB/s written; copy: {c:.1f} us = {2*mb/c:.2f} TB/s read+written")   # MB/us = TB/s

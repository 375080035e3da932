"""Known-byte-count kernels for calibrating FETCH_SIZE / WRITE_SIZE (run under rocprofv3 --pmc):
zero_ of 1 GiB (1 GiB written, nothing read) and copy_ of 1 GiB (1 GiB read + 1 GiB written)."""
import torch

n = 1 << 28  # 2^28 float32 = 1 GiB
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    x.zero_()
    torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)
    torch.cuda.synchronize()
print("calibration done: zero_ x3 (1 GiB write each), copy_ x3 (1 GiB read + 1 GiB write each)")

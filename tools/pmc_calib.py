"""PMC calibration on a known byte count (MI355X_MICROARCH.md, HBM): a 512 MiB
device-to-device copy kernel with 16 B/lane accesses; FETCH_SIZE / WRITE_SIZE
are then read for it next to our kernels."""
import torch
x = torch.empty(512 * 1024 * 1024 // 4, dtype=torch.int32, device="cuda").fill_(3)
y = torch.empty_like(x)
torch.cuda.synchronize()
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
print("copied", x.numel() * 4, "bytes x3")

"""Practical HBM rates on this GPU (calibration for the roofline discussion in DESIGN.md): device-to-device
copy, fill (write only) and a sum reduction (read only) over 1 GiB buffers, best of 5."""
import torch, time
x = torch.empty(1 << 28, dtype=torch.int32, device="cuda"); y = torch.empty_like(x); x.fill_(1)
def best(fn, reps=5):
    t = []
    for _ in range(reps):
        torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); t.append(a.elapsed_time(b))
    return min(t)
nb = x.numel() * 4
print("copy  (read+write) %.2f TB/s" % (2 * nb / best(lambda: y.copy_(x)) / 1e9))
print("fill  (write)      %.2f TB/s" % (nb / best(lambda: y.fill_(7)) / 1e9))
print("sum   (read)       %.2f TB/s" % (nb / best(lambda: x.sum()) / 1e9))

#!/usr/bin/env python3
"""What HBM bandwidth a plain streaming kernel reaches on this box, for K1's read : write mix (calibration of the 8 TB/s
roofline used in bench.py): copies / scales of f64 arrays the size of C2's particle attributes (67 M x 8 B = 537 MB)."""
import json, time, torch
dev = torch.device("cuda:0")
n = 1 << 26
a = [torch.rand(n, device=dev, dtype=torch.float64) for _ in range(8)]
b = [torch.empty(n, device=dev, dtype=torch.float64) for _ in range(7)]
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
out = {}
t = timed(lambda: [b[i].copy_(a[i]) for i in range(7)]); out["copy_7_arrays_TBps"] = 7 * 2 * n * 8 / t / 1e12
t = timed(lambda: [torch.mul(a[i], 1.0001, out=b[i]) for i in range(7)]); out["scale_7_arrays_TBps"] = 7 * 2 * n * 8 / t / 1e12
t = timed(lambda: [a[i].sum() for i in range(8)]); out["read_only_8_arrays_TBps"] = 8 * n * 8 / t / 1e12
t = timed(lambda: [b[i].fill_(1.0) for i in range(7)]); out["write_only_7_arrays_TBps"] = 7 * n * 8 / t / 1e12
# one fused kernel with K1's mix: 8 arrays read, 7 written
t = timed(lambda: torch._foreach_mul_(b, 1.0001)); out["foreach_inplace_rw_7_arrays_TBps"] = 7 * 2 * n * 8 / t / 1e12
print(json.dumps(out))

"""Infinity-Cache probe: in-place read+write sweeps and write->read pairs over buffers of growing size.
Run on the GPU box: python tools/mall_probe.py"""
import torch, json
dev = torch.device("cuda:0")
def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
out = []
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    t_inplace = timed(lambda: x.mul_(1.0000001), 30)
    t_copy = timed(lambda: y.copy_(x), 30)
    t_read = timed(lambda: x.sum(), 30)
    def pair():
        y.copy_(x)          # write y (and read x)
        y.sum()             # read y back
    t_pair = timed(pair, 30)
    rec = dict(mb=mb, inplace_gbps=2 * n * 4 / t_inplace / 1e9, copy_gbps=2 * n * 4 / t_copy / 1e9,
               read_gbps=n * 4 / t_read / 1e9, copy_then_sum_us=t_pair * 1e6,
               copy_us=t_copy * 1e6, sum_us=t_read * 1e6)
    out.append(rec); print(json.dumps(rec), flush=True)

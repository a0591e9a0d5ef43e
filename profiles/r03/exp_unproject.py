"""unproject (depth_to_points for whole frames): store flavour and the chip's write ceiling.
usage: LA3D_LIB=<lib> python profiles/r03/exp_unproject.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import labelany3d_amd as la

dev = torch.device("cuda", 0)
H, W = 480, 640


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


K = torch.tensor([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]], dtype=torch.float64, device=dev)
for P in (64, 256, 1024):
    stack = torch.rand(P, H, W, device=dev) * 9 + 0.5
    Kst = K[None].expand(P, 3, 3).contiguous()
    t = timed(lambda: la.unproject(stack, Kst))
    print(f"{os.environ.get('LA3D_LIB','product')}: unproject f64 P={P}: {t*1e6:8.1f} us  {P*H*W*28/t/1e9:7.1f} GB/s")
    buf = torch.empty(P * H * W * 3, dtype=torch.float64, device=dev)
    t = timed(lambda: buf.zero_())
    print(f"   zero_() of the same output: {t*1e6:8.1f} us  {buf.numel()*8/t/1e9:7.1f} GB/s (pure write)")
    src = torch.empty_like(buf)
    t = timed(lambda: buf.copy_(src))
    print(f"   copy_() same size: {t*1e6:8.1f} us  {2*buf.numel()*8/t/1e9:7.1f} GB/s (read+write)")
    del stack, buf, src

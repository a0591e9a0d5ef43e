"""depth_to_points for a stack of frames (la3d_unproject_batch), raw C calls on preallocated outputs, HIP events: GB/s by batch size,
output type and frame size (28 B per pixel for f64 out: 4 read + 24 written; 16 B for f32 out)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from labelany3d_amd._lib import lib, check

dev = torch.device("cuda", 0)
K1 = np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]])


def run(P, H, W, f64, n=20):
    depth = torch.rand((P, H, W), device=dev) * 9.5 + 0.5
    K = torch.as_tensor(np.repeat(K1[None], P, 0), device=dev)
    out = torch.empty((P, H * W, 3), dtype=torch.float64 if f64 else torch.float32, device=dev)
    s = torch.cuda.current_stream()
    call = lambda: check(lib.la3d_unproject_batch(depth.data_ptr(), K.data_ptr(), 9, None, P, H, W, out.data_ptr(), int(f64), s.cuda_stream), "unproject")
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n * 1e-3
    by = P * H * W * (28 if f64 else 16)
    print(f"P={P:5d} {H}x{W} out={'f64' if f64 else 'f32'}: {t * 1e6:8.1f} us  {by / t / 1e9:7.0f} GB/s  {P / t:10.0f} frames/s")


for P in (1, 8, 64, 256, 1024):
    run(P, 480, 640, True)
for P in (64, 256):
    run(P, 480, 640, False)
run(1, 2160, 3840, True)
run(8, 2160, 3840, True)

import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import labelany3d_amd as la
dev = torch.device("cuda", 0)
def timed(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for H, W in ((640, 427), (375, 500)):
    for B in (1, 8, 64, 256, 1024):
        rs = np.random.RandomState(B)
        depth = torch.rand((H, W), device=dev) * 9 + 0.5
        K = torch.tensor([[500.0, 0, W / 2], [0, 500.0, H / 2], [0, 0, 1]], dtype=torch.float64, device=dev)
        m = np.zeros((B, H, W), np.uint8)
        for i in range(B):
            h, w = rs.randint(8, int(0.6 * H)), rs.randint(8, int(0.5 * W))
            r0, c0 = rs.randint(0, H - h + 1), rs.randint(0, W - w + 1)
            m[i, r0:r0 + h, c0:c0 + w] = 1
        masks = torch.as_tensor(m, device=dev)
        ii = torch.zeros(B, dtype=torch.int32, device=dev)
        t_plain = timed(lambda: la.fit_instances(depth, masks, K, image_index=ii))
        Wp = la.padded_width(W)
        def padded():
            mp = torch.nn.functional.pad(masks, (0, Wp - W))
            dp, _ = la.pad_depth_rows(depth)
            return la.fit_instances(dp, mp, K, image_index=ii)
        t_pad = timed(padded)
        mp = torch.nn.functional.pad(masks, (0, Wp - W)); dp, _ = la.pad_depth_rows(depth)
        t_pre = timed(lambda: la.fit_instances(dp, mp, K, image_index=ii))
        a = la.fit_instances(depth, masks, K, image_index=ii)[0]; b = padded()[0]
        err = float((a - b).abs().max())
        print(f"{H}x{W} B={B:5d}: as given {t_plain:7.1f} us | padded per call {t_pad:7.1f} | padded once {t_pre:7.1f} | max |diff| {err:.1e}", flush=True)

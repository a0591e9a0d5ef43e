"""Per-step HIP-event times right after a synchronize(): is the first ~2 ms of a timed region slower than the steady state?
(bench.py reads 113 / 109 / 107.7 us per step at K = 20 / 200 / 1000.)  usage: python profiles/r03/exp_step_ramp.py [idle_ms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from labelany3d_amd import InstanceFitter

dev = torch.device("cuda", 0)
depth, masks, K, n_masked, rects = bench.make_inputs(1024, dev, 1234)
fit = InstanceFitter(1024, bench.H, bench.W, dev, slots=1, ws_slots=2)
st = torch.cuda.current_stream()
for idle_ms in [0.0, 1.0, 20.0, 500.0]:
    for warm in [5, 200]:
        for _ in range(warm):
            fit.run(depth, masks, K, slot=0, stream=st)
        torch.cuda.synchronize()
        time.sleep(idle_ms * 1e-3)
        N = 60
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
        ev[0].record(st)
        for k in range(N):
            fit.run(depth, masks, K, slot=0, stream=st)
            ev[k + 1].record(st)
        torch.cuda.synchronize()
        t = np.array([ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(N)])
        print(f"idle {idle_ms:6.1f} ms warm {warm:4d}: steps 0-4 {np.round(t[:5],1)} | 5-19 mean {t[5:20].mean():.1f} | 20-39 {t[20:40].mean():.1f} | 40-59 {t[40:].mean():.1f}")

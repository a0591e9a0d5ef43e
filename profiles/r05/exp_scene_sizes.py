"""The scene pipeline on a COCO-like MIX of frame sizes (batches are formed per size; odd widths run on padded rows): images/s for
2048 in-memory scenes of one size against the same number drawn from the sizes COCO val2017 comes in."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from labelany3d_amd import fit_scenes as F

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
# (height, width, share) - the most frequent sizes of COCO val2017 and a tail of rare ones
SIZES = [(480, 640, .25), (427, 640, .20), (640, 480, .07), (426, 640, .06), (375, 500, .06), (640, 427, .05), (428, 640, .03), (333, 500, .03),
         (500, 375, .03), (360, 640, .02), (612, 612, .02), (425, 640, .02), (640, 426, .02), (424, 640, .015), (334, 500, .015), (512, 640, .01),
         (500, 333, .01), (640, 428, .01), (480, 480, .01), (376, 500, .01), (457, 640, .01), (400, 600, .005), (431, 640, .005), (640, 512, .005)]


def run(scenes, tag):
    best = None
    for rep in range(3):
        t = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = sum(len(r) for _, r in F.ScenePipeline(batch_images=256, write=False, timings=t).run(scenes))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{tag}: {len(scenes)} images, {n} boxes, {len(scenes) / best:8.0f} images/s (best of 3), {int(t['batches'])} batches", flush=True)


uni, _ = F.synthetic_scenes(N, seed=1)
run(uni, "640x480 only        ")
rs = np.random.RandomState(0)
share = np.array([s[2] for s in SIZES]); share /= share.sum()
counts = rs.multinomial(N, share)
mix = []
for (H, W, _), c in zip(SIZES, counts):
    if c:
        sc, _ = F.synthetic_scenes(int(c), seed=H * 1000 + W, H=H, W=W)
        for k, s in enumerate(sc):
            s["name"] = f"{H}x{W}_{k}"
        mix += sc
order = rs.permutation(len(mix))
mix = [mix[i] for i in order]
run(mix, f"COCO-like mix ({int((counts > 0).sum())} sizes)")

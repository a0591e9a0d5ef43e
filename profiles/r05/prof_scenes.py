"""cProfile of the host side of the real-data pipeline (bench.py --end-to-end): where the Python time of ScenePipeline goes, per thread
(the producer packs and uploads, the caller issues the fit calls and formats the records)."""
import cProfile
import io
import os
import pstats
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from labelany3d_amd.fit_scenes import ScenePipeline, synthetic_scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
scenes, _ = synthetic_scenes(n, seed=3)
dev = torch.device("cuda", 0)
list(ScenePipeline(device=dev, batch_images=256, write=False).run(scenes[:512]))
profs = {}
orig = threading.Thread.run


def run(self, *a, **k):       # profile every thread separately (cProfile is per thread)
    pr = cProfile.Profile()
    profs[self.name] = pr
    pr.enable()
    try:
        return orig(self, *a, **k)
    finally:
        pr.disable()


threading.Thread.run = run
main = cProfile.Profile()
main.enable()
list(ScenePipeline(device=dev, batch_images=256, write=False).run(scenes))
main.disable()
profs["main"] = main
for name, pr in profs.items():
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
    txt = s.getvalue()
    if "function calls" in txt:
        print(f"===== thread {name}\n" + "\n".join(txt.splitlines()[:48]))

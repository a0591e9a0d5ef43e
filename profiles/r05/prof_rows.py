import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench, labelany3d_amd as la
from labelany3d_amd.options import scheduling
B = int(sys.argv[1]); eng = sys.argv[2]
dev = torch.device("cuda", 0)
depth, masks, K, n_masked, rects = bench.make_inputs(B, dev, 1234)
f = la.InstanceFitter(B, bench.H, bench.W, dev)
st = torch.cuda.current_stream()
for _ in range(300):
    f.run(depth, masks, K, engine=eng, stream=st)
torch.cuda.synchronize()

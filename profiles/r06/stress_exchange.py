"""Round 6: a soak test of the engines whose workgroups hand data to each other INSIDE one launch (band engine: 2 / 4 / 8 workgroups
per instance meeting through the workspace for the moments and the extents; row engine: up to 16 bands per instance, the band that
arrives last merges) - the code where a memory-ordering or tag mistake would show as a rare wrong record, not as a failing unit test.

    python profiles/r06/stress_exchange.py [--calls 20000] [--out profiles/r06/stress_exchange.txt]

Per configuration (batch size x engine x ground): three resident input sets; the expected records of each = the engine's own first
call, checked against the instance engine (one workgroup per instance, no exchange) at 1e-11.  Then --calls calls back to back on one
stream cycling through the input sets and three output slots, and the same number split over two streams with a workspace each
(calls of different batches in flight at the same time); EVERY call's records, status and aux are compared bit for bit with the
expected ones on the device (a mismatch counter, read once at the end)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=20000)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06", "stress_exchange.txt"))
    a = ap.parse_args()

    import torch

    import bench
    from labelany3d_amd import InstanceFitter

    dev = torch.device("cuda", 0)
    H, W = bench.H, bench.W
    lines, total_bad, total_calls = [], 0, 0
    t_all = time.time()
    for B, engine, grounded in [(1, "band", True), (8, "band", True), (64, "band", True), (128, "band", True), (144, "band", True), (256, "band", True),
                                (1, "rows", False), (16, "rows", False), (64, "rows", False), (128, "rows", False), (160, "rows", False),
                                (8, "band", False), (37, "rows", False)]:
        sets = []
        for k in range(3):
            depth, masks, K, _, _ = bench.make_inputs(B, dev, 100 + 17 * k + B)
            g = None
            if grounded:
                g = torch.as_tensor(np.array([[0.05, -0.97, 0.1, 1.2]] * B) + 0.02 * np.random.RandomState(B + k).randn(B, 4), device=dev)
            sets.append((depth, masks, K, g))
        f = InstanceFitter(B, H, W, dev, slots=3, ws_slots=2)
        exp = []
        for depth, masks, K, g in sets:
            b, s, x = (t.clone() for t in f.run(depth, masks, K, ground=g, slot=0, engine=engine))
            bi, si, _ = (t.clone() for t in f.run(depth, masks, K, ground=g, slot=1, engine="instance"))
            torch.cuda.synchronize()
            assert torch.equal(s, si) and int((s == 0).sum()) > 0
            ok = s == 0
            torch.testing.assert_close(b[ok][:, :15], bi[ok][:, :15], rtol=1e-11, atol=1e-11)
            exp.append((b, s, x))
        bad = torch.zeros((), dtype=torch.int64, device=dev)

        def check(k, slot):
            nonlocal bad
            b, s, x = f.boxes[slot], f.status[slot], f.aux[slot]
            eb, es, ex = exp[k]
            # (bit patterns: NaN records of rejected instances compare equal to themselves)
            bad += (b.view(torch.int64) != eb.view(torch.int64)).any().long() + (s != es).any().long() + (x.view(torch.int64) != ex.view(torch.int64)).any().long()

        t0 = time.time()
        for i in range(a.calls):                       # one stream, back to back
            k, slot = i % 3, i % 3
            depth, masks, K, g = sets[k]
            f.run(depth, masks, K, ground=g, slot=slot, engine=engine)
            check(k, slot)
        torch.cuda.synchronize()
        t1 = time.time()
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        f2 = [InstanceFitter(B, H, W, dev, slots=1) for _ in range(2)]
        bad2 = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(2)]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        for i in range(a.calls // 2):                  # two streams at once, a fitter (workspace) each
            for j, st in enumerate(streams):
                k = (i + j) % 3
                depth, masks, K, g = sets[k]
                with torch.cuda.stream(st):
                    f2[j].run(depth, masks, K, ground=g, slot=0, stream=st, engine=engine)
                    eb, es, ex = exp[k]
                    bad2[j] += (f2[j].boxes[0].view(torch.int64) != eb.view(torch.int64)).any().long() + (f2[j].status[0] != es).any().long()
        torch.cuda.synchronize()
        nb = int(bad.item()) + sum(int(x.item()) for x in bad2)
        total_bad += nb
        total_calls += a.calls + 2 * (a.calls // 2)
        lines.append(f"B = {B:4d} {engine:5s} ground = {str(grounded):5s}: {a.calls} calls on one stream ({(t1 - t0) / a.calls * 1e6:.1f} us per call incl. the comparison) + "
                     f"{2 * (a.calls // 2)} on two streams: {nb} mismatching calls")
        print(lines[-1], flush=True)
    lines.append(f"total: {total_calls} calls, {total_bad} mismatches, {time.time() - t_all:.0f} s")
    print(lines[-1])
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()

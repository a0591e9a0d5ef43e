"""The measurement harness itself: the all-cores CPU-baseline worker protocol (CPU) and the JSON line bench.py prints
(GPU)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_worker_protocol(tmp_path):
    rs = np.random.RandomState(0)
    d = rs.uniform(0.5, 10, (3, 48, 64)).astype(np.float32)
    m = np.zeros((3, 48, 64), np.uint8)
    m[:, 10:30, 20:50] = 1
    np.save(tmp_path / "d.npy", d)
    np.save(tmp_path / "m.npy", m)
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", str(tmp_path / "d.npy"), str(tmp_path / "m.npy"), "1", "4"],
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    assert p.stdout.readline().startswith("ready")
    p.stdin.write("go\n")
    p.stdin.flush()
    line = p.stdout.readline()
    p.wait(timeout=60)
    assert line.startswith("done") and float(line.split()[1]) > 0


@pytest.mark.gpu
def test_bench_prints_one_contract_line():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["metric"] == "fitted 3D boxes/sec @640x480" and d["unit"] == "boxes/s" and d["n_gpus"] == 1 and d["steps"] == 5
    assert d["vs_baseline"] is None and d["scaling"] == "weak" and d["higher_is_better"] is True and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.0 < r["frac"] < 1.0, r["frac"]                      # a fraction of a physical roof
    assert r["required_bytes_per_launch"] < r["algorithmic_bytes_per_launch"] and "byte_model" in r
    assert abs(r["achieved"] - r["required_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic_stale"] in (True, False, None)
    v = r["valu"]                                                 # the second roof (profiles/valu_per_launch.json), optional
    assert v is None or (0.0 < v["frac_of_step"] < 1.0 and v["stale"] in (True, False) and v["wave_instructions_per_step"] > 1e6)
    assert abs(d["value"] - 5 * 1024 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-6
    assert d["value"] > 2.08e6   # BASELINE target: 40 % of the HBM-read roofline
    # round 6: the timed steps rotate through distinct resident input batches; the line says which binary produced it
    rot = d["rotation"]
    assert rot["batches"] == 3 and d["config"]["input_batches"] == 3 and len(rot["required_bytes_per_batch"]) == 3
    assert len(set(rot["required_bytes_per_batch"])) == 3                    # different masks, same distribution
    mean_req = sum(rot["required_bytes_per_batch"][k % 3] for k in range(5)) / 5
    assert abs(r["required_bytes_per_launch"] - mean_req) < 1e-6 * mean_req
    assert rot["same_batch_ms_per_step"] > 0 and abs(rot["same_batch_ms_per_step"] / r["avg_launch_ms"] - 1) < 0.25
    b = d["build"]
    assert len(b["lib_sources_sha256"]) == 64 and len(b["lib_compile_cmd_sha256"]) == 64 and b["lib_built_from_tree"] is True
    assert b["kernel_source_sha256"] == __import__("bench").kernel_source_sha256()


@pytest.mark.gpu
def test_bench_rotate_1_is_the_old_protocol_and_cpu_baseline_names_the_reference_probe():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--rotate", "1", "--no-steady",
                          "--no-pipelined", "--batch", "256"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["rotation"]["batches"] == 1 and d["rotation"]["same_batch_ms_per_step"] is None
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    assert cb["reference_probe"]["value"] == 573.0 and cb["reference_probe"]["unit"] == "boxes/s/core"


def test_required_bytes_model():
    """bench.required_bytes: mask planes once + 1 KiB per 32x8 tile that holds a mask pixel + records (CPU tensors)."""
    import torch

    import bench

    m = torch.zeros((3, 480, 640), dtype=torch.uint8)
    m[0, 0, 0] = 1                       # one tile
    m[1, 7:9, 31:33] = 1                 # straddles 2x2 tiles
    req, tiles = bench.required_bytes(m)  # plane 2 empty
    assert tiles == 5 and req == 3 * 480 * 640 + 5 * 1024 + 3 * 312
    m2 = torch.ones((1, 50, 70), dtype=torch.uint8)   # ragged frame: ceil(50/8) x ceil(70/32) tiles
    assert bench.required_bytes(m2)[1] == 7 * 3
    assert len(bench.kernel_source_sha256()) == 64


def test_bench_refuses_to_measure_fewer_gpus_than_asked():
    """`python bench.py --gpus N` on a node with fewer than N GPUs (this container: none; the one-GPU box: one) must fail
    loudly instead of recording a one-GPU number under n_gpus = N - both as its own launcher and under a torchrun line whose
    world size disagrees with --gpus."""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this node could really run two ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LA3D_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 2 and "needs 2 GPUs" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_plain_bench_command_launches_its_own_ranks():
    """The shape of the driver's N = 1 command with --gpus 2 - no torch.distributed.run in front - must come back as a TWO-rank
    line: bench.py re-runs itself under the launcher (here: gloo dry run, both ranks on this box's one GPU) and stamps the
    ranks it really ran into the line."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LA3D_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "512",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4
    rk = d["ranks"]
    assert rk["world_size"] == 2 and rk["backend"] == "gloo" and rk["rccl_ranks"] == 0 and rk["self_launched"] is True
    assert [e["rank"] for e in rk["devices"]] == [0, 1] and len({e["pid"] for e in rk["devices"]}) == 2
    assert all(e["name"] for e in rk["devices"])


@pytest.mark.gpu
def test_single_gpu_line_names_its_device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--no-steady", "--no-pipelined"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    rk = d["ranks"]
    assert d["n_gpus"] == 1 and rk["world_size"] == 1 and rk["backend"] is None and rk["self_launched"] is False
    assert len(rk["devices"]) == 1 and rk["devices"][0]["device"] == 0

"""bench.py's N > 1 control flow on CPU: two gloo ranks run bench.main() with the device layer replaced by stand-ins
(torch.cuda calls, the NCCL process group and xrsfm_amd.capi.Context).  What is under test is the launcher contract: rank /
world handling, per-rank shard generation (weak) or sharding (strong), the size all-reduce, the unique-id broadcast, the
barrier + MAX-over-ranks timing and the single JSON line of rank 0.  The numerics are the GPU tests' business."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import os, sys, types
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a: None
_tensor, _zeros = torch.tensor, torch.zeros
torch.tensor = lambda *a, **k: _tensor(*a, **{{x: y for x, y in k.items() if x != "device"}})
torch.zeros = lambda *a, **k: _zeros(*a, **{{x: y for x, y in k.items() if x != "device"}})
_init = dist.init_process_group
dist.init_process_group = lambda backend=None, device_id=None, **k: _init(backend="gloo", **k)
from xrsfm_amd import capi, synth
synth.CONFIGS["S"] = dict(n_cams=12, n_points=300, k_obs=4, seed=2)          # keep the generated problems tiny
synth.CONFIGS["T"] = dict(n_cams=60, n_points=400, seed=12, cams_per_cluster=20)      # (BASELINE config 5's generator, at toy size)
LOG = []
class FakeSummary:
    n_successful, n_unsuccessful, pcg_iterations, linear_solver_used, termination_reason = 5, 1, 0, 1, 3
    initial_cost, final_cost = 10.0, 1.0
class FakeContext:
    def __init__(self, prob, device=0): self.prob = prob; LOG.append(("ctx", device, prob.n_points, prob.n_obs))
    def comm_init(self, world, rank, uid): LOG.append(("comm", world, rank, len(uid), sum(uid)))
    def reset(self): pass
    def run(self, opt=None): return FakeSummary()
    def download(self): return self.prob.cam_q, self.prob.cam_t, self.prob.points
    def profile(self): return {{"k_schur_pairs": (1.4, 14), "k_linearize": (1.2, 15)}}
    def close(self): pass
capi.device_count = lambda: 1
capi.Context = FakeContext
capi.comm_unique_id = lambda: bytes(range(128))
import bench
sys.argv = ["bench.py"] + {argv!r}
bench.main()
if os.environ.get("DRYRUN_LOG_DIR"):          # (under one launcher the ranks share a pipe: their lines may run into each other)
    open(os.path.join(os.environ["DRYRUN_LOG_DIR"], "rank%s.log" % os.environ.get("RANK", "0")), "w").write(repr(LOG))
else:
    print("LOG", LOG)
'''


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_torchrun_launch_line_of_the_driver(tmp_path):
    """The driver's own launch line for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...`), with the stand-in device layer: both ranks come up, rendezvous on
    127.0.0.1, and exactly one JSON line leaves the job."""
    port = _free_port()
    argv = ["--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "S", "--no-cpu"]
    script = tmp_path / "bench_standin.py"
    script.write_text(DRIVER.format(root=ROOT, argv=argv))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DRYRUN_LOG_DIR"] = str(tmp_path)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and "sharded x2" in j["config"]["parallelism"]
    logs = [eval(open(os.path.join(str(tmp_path), f"rank{r}.log")).read()) for r in range(2)]
    assert sorted(log[1][2] for log in logs) == [0, 1] and all(log[1][:2] == ("comm", 2) and log[1][3:] == (128, sum(range(128))) for log in logs)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_rank_launch_contract(tmp_path, scaling):
    port = _free_port()
    argv = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "S", "--no-cpu", "--scaling", scaling]
    script = tmp_path / "driver.py"
    script.write_text(DRIVER.format(root=ROOT, argv=argv))
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    lines1 = [l for l in outs[1][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and len(lines1) == 0                      # ONE JSON line, from rank 0
    j = json.loads(lines0[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == scaling and j["higher_is_better"] is True
    assert j["unit"] == "cam-pts*iter/s" and j["dtype"] == "f64" and j["data"] == "synthetic" and j["vs_baseline"] is None
    n_points_total = 600 if scaling == "weak" else 300                # weak: every rank holds a config-sized shard
    assert f"/ {n_points_total} points /" in j["config"]["workload"] and "sharded x2" in j["config"]["parallelism"]
    assert abs(j["value"] - 12 * (12 + n_points_total) / (j["ms_per_step"] * 2e-3)) <= 1e-6 * j["value"]     # 2 steps x 6 LM iterations
    assert j["roofline"]["kernel"] == "k_schur_pairs" and j["cpu_baseline"] is None
    # every rank: its own device, its own shard, the same communicator id
    logs = [eval([l for l in o.splitlines() if l.startswith("LOG")][0][4:]) for o, _ in outs]
    per_rank_points = 300 if scaling == "weak" else 150
    for rank, log in enumerate(logs):
        assert log[0][:3] == ("ctx", rank, per_rank_points)
        assert log[1] == ("comm", 2, rank, 128, sum(range(128)))


def test_two_ranks_on_the_collection_config(tmp_path):
    """`bench.py --gpus 2 --config T` (BASELINE config 5: the unordered photo collection, the configuration BASELINE.json places on
    8 GPUs): strong scaling — the collection is generated once and its tracks are sharded over the ranks by the length-aware
    partition, cameras replicated; one JSON line, the workload named as the clustered collection."""
    port = _free_port()
    argv = ["--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "T", "--no-cpu", "--no-extras"]
    script = tmp_path / "driver.py"
    script.write_text(DRIVER.format(root=ROOT, argv=argv))
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    lines0 = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines0) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    j = json.loads(lines0[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and "clustered photo collection" in j["config"]["workload"] and "sharded x2" in j["config"]["parallelism"]
    logs = [eval([l for l in o.splitlines() if l.startswith("LOG")][0][4:]) for o, _ in outs]
    pts = [log[0][2] for log in logs]
    obs = [log[0][3] for log in logs]
    assert abs(obs[0] - obs[1]) <= 0.02 * sum(obs) and min(pts) > 0            # observations balanced within 1 % of the total per rank
    assert all(log[1] == ("comm", 2, rank, 128, sum(range(128))) for rank, log in enumerate(logs))

"""CPU: the N > 1 path - env sharding + the single observation all-gather - with world_size 2 over gloo."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import REPO

WORKER = textwrap.dedent(
    """
    import os, sys, torch
    sys.path.insert(0, sys.argv[1])
    import torch.distributed as dist
    from tacex_amd.env_shard import ObservationGather, init_from_env
    shard = init_from_env(num_envs_total=6, backend="gloo")
    assert shard.world_size == 2 and shard.num_local == 3 and (shard.lo, shard.hi) == (3 * shard.rank, 3 * shard.rank + 3)
    g = ObservationGather({"rgb32": (2, 2, 3), "indent": (1,)}, shard.num_local, shard.world_size, "cpu")
    env_ids = torch.arange(shard.lo, shard.hi, dtype=torch.float32)
    g.pack("rgb32", env_ids.view(-1, 1, 1, 1).expand(-1, 2, 2, 3))
    g.pack("indent", env_ids * 10)
    out = g.gather()   # exactly one collective
    assert out["indent"].reshape(-1).tolist() == [0., 10., 20., 30., 40., 50.], out["indent"]
    assert out["rgb32"][:, 0, 0, 0].tolist() == [0., 1., 2., 3., 4., 5.]
    # the pipelined form bench.py uses (mixed dtypes, one byte buffer): issue, refill after the implicit wait, read views
    g2 = ObservationGather({"rgb32": (2, 2, 3), "indent": (1,)}, shard.num_local, shard.world_size, "cpu",
                           dtypes={"rgb32": torch.uint8})
    for step in range(3):
        g2.pack_all({"rgb32": (env_ids + step).to(torch.uint8).view(-1, 1, 1, 1).expand(-1, 2, 2, 3), "indent": env_ids * 10 + step})
        g2.gather_async()
    v = g2.views()
    assert v["rgb32"].dtype == torch.uint8 and v["rgb32"][:, 1, 1, 2].tolist() == [2, 3, 4, 5, 6, 7]
    assert v["indent"].reshape(-1).tolist() == [2., 12., 22., 32., 42., 52.]
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + shard.rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == 2.0
    dist.barrier()
    dist.destroy_process_group()
    print("rank", shard.rank, "ok")
    """
)


def test_two_rank_gather_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(REPO)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{o}"
        assert f"rank {rank} ok" in o


def test_bench_gpus_2_starts_its_own_ranks_and_prints_one_compact_line(tmp_path):
    """`python bench.py --gpus 2` with NO rendezvous in the environment (how the driver invokes it): the parent starts two fresh child
    ranks itself, they meet over gloo (--cpu-dry-run: the launcher / barrier / max-over-ranks / all-gather / emitter path with a
    stand-in step), and stdout is exactly ONE JSON line within the driver's 8 KB tail."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    det = tmp_path / "details.json"
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--cpu-dry-run",
                        "--details-out", str(det)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) <= 6000, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["data"] == "dry-run" and d["scaling"] == "weak"
    m = d["multi_gpu"]
    assert m["world_size"] == 2 and m["backend"] == "gloo" and m["launcher"] == "self" and len(m["per_rank_ms_per_step"]) == 2
    assert abs(d["ms_per_step"] - max(m["per_rank_ms_per_step"])) < 1e-3  # MAX over ranks
    assert abs(d["value"] - d["config"]["frames_per_step"] / (d["ms_per_step"] * 1e-3)) <= 2e-2 * d["value"]
    assert d["config"]["frames_per_step"] == 2 * 1024 * 2
    assert json.loads(det.read_text())["multi_gpu"]["rank_devices"] == ["cpu-0", "cpu-1"]
    # the north-star point (4096 envs over the whole node, 2048 per rank here) rides on the same line: with / without the gather, C4-shaped
    # with / without it, and the one-GPU base of the two jobs (rank 0 alone, no collective, the other rank waits at the barrier)
    for k in ("value_node4096", "value_node4096_no_gather", "value_node4096_fem", "value_node4096_fem_no_gather"):
        assert d[k] > 0, k
    assert set(d["strong_scaling_base"]) == {"node4096", "node4096_fem"} and all(v > 0 for v in d["strong_scaling_base"].values())
    nd = json.loads(det.read_text())["node4096"]
    assert nd["envs_total"] == 4096 and nd["envs_per_gpu"] == 2048


def test_bench_under_torchrun_contract_env(tmp_path):
    """The other launch form (`python -m torch.distributed.run ... bench.py --gpus 2`): RANK / WORLD_SIZE come from the environment,
    bench.py must not start children of its own."""
    import json

    port = 29500 + (os.getpid() % 2000) + 7
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-dry-run",
                                       "--details-out", str(tmp_path / f"d{rank}.json")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    assert outs[1][0].strip() == ""  # only rank 0 prints
    d = json.loads(outs[0][0].strip())
    assert d["multi_gpu"]["launcher"] == "torch.distributed.run" and d["n_gpus"] == 2


@pytest.mark.gpu
def test_rccl_observation_gather_single_rank_child_process(tmp_path):
    """RCCL smoke on the GPU box (the 8-GPU run must not also be the first RCCL run): a FRESH child process joins a 1-rank
    `nccl` (= RCCL) group and drives five bench-style steps through ObservationGather.pack_all() / gather_async(); the
    gathered views must equal the packed inputs, including the mixed uint8 / float32 byte layout."""
    import os
    import socket
    import subprocess
    import sys

    from conftest import REPO

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "rccl_smoke.py"
    script.write_text(
        "import os, sys, torch\n"
        f"sys.path.insert(0, {str(REPO)!r})\n"
        "import torch.distributed as dist\n"
        "from bench import Rig\n"
        "from tacex_amd.env_shard import init_from_env\n"
        "shard = init_from_env(8, backend='nccl')\n"
        "assert dist.is_initialized() and dist.get_backend() == 'nccl' and shard.world_size == 1\n"
        "torch.cuda.set_device(0)\n"
        "rig = Rig(8, 240, 320, 2, True, 'cuda:0', shard.world_size, seed=3)\n"
        "assert not rig.obs._alias  # a real collective fills `full`, not an alias of the send buffer\n"
        "for i in range(5):\n"
        "    rig.step(i)\n"
        "rig.finish()\n"
        "torch.cuda.synchronize()\n"
        "v = rig.obs.views()\n"
        "for k, s in enumerate(rig.sensors):\n"
        "    o = s._data.output\n"
        "    assert torch.equal(v[f'rgb32_{k}'], o['tactile_rgb_obs']), 'rgb32'\n"
        "    assert v[f'rgb32_{k}'].dtype == torch.uint8 and float(v[f'rgb32_{k}'].float().mean()) > 1.0\n"
        "    assert torch.equal(v[f'markers_{k}'], o['marker_motion']), 'markers'\n"
        "    assert torch.equal(v[f'indent_{k}'][:, 0], s.indentation_depth), 'indent'\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "print('RCCL_SMOKE_OK')\n")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               TACEX_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SMOKE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.gpu
def test_bench_self_launch_with_rccl_on_the_gpu_box(tmp_path):
    """`python bench.py --gpus N` as the driver invokes it - no rendezvous in the environment - on real hardware: the parent (which never
    touches the GPU) starts the rank(s) as fresh child processes, they form an RCCL (`nccl`) group, run the headline rig with the
    observation all-gather, and the parent relays rank 0's ONE line.  One device here, so N = 1 through TACEX_BENCH_FORCE_LAUNCH."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(TACEX_BENCH_FORCE_LAUNCH="1", TACEX_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--envs-per-gpu", "64", "--no-sweep",
                        "--no-cpu-baseline", "--details-out", str(tmp_path / "d.json")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) <= 6000, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["data"] == "synthetic" and d["value"] > 0 and "roofline" in d
    m = d["multi_gpu"]
    assert m["backend"] == "nccl" and m["world_size"] == 1 and m["launcher"] == "self" and m["distinct_devices"] == 1

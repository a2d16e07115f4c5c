"""CPU tests of bench.py's launcher-less start (`python bench.py --gpus N`, the command the driver runs): the rank spawner
itself with a stub rank script over gloo, and the refusal — one JSON line, non-zero exit, no hang — when the host has
fewer GPUs than RCCL ranks were asked for.  The GPU leg is tests/test_gpu_multi.py."""
import json
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
CLEAN = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}


def test_spawn_ranks_starts_world_and_relays_rank0_line(tmp_path):
    stub = tmp_path / "rank_stub.py"
    stub.write_text(textwrap.dedent("""
        import json, os, sys
        import torch, torch.distributed as dist
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        v = torch.tensor([float(os.environ["RANK"]) + 1.0], dtype=torch.float64)
        dist.all_reduce(v)
        if dist.get_rank() == 0:
            print(json.dumps({"world": dist.get_world_size(), "sum": float(v.item()), "argv": sys.argv[1:],
                              "addr": os.environ["MASTER_ADDR"]}))
        dist.barrier(); dist.destroy_process_group()
    """))
    drv = tmp_path / "drv.py"
    drv.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        raise SystemExit(bench.spawn_ranks(3, {str(stub)!r}, ["--gpus", "3", "--steps", "2"], timeout_s=240))
    """))
    r = subprocess.run([sys.executable, str(drv)], env=CLEAN, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"world": 3, "sum": 6.0, "argv": ["--gpus", "3", "--steps", "2"], "addr": "127.0.0.1"}


def test_spawn_ranks_at_the_drivers_world_size(tmp_path):
    """Eight ranks — `python bench.py --gpus 8`, the driver's largest run — through the same spawner: rendezvous on 127.0.0.1 with a port the
    launcher picks itself, the division of the host's threads over the ranks (OMP_NUM_THREADS), an all-gather of one slice per rank in rank
    order (the farm's only exchange) and a MAX-reduced time, as bench.py's loop does them."""
    stub = tmp_path / "rank8_stub.py"
    stub.write_text(textwrap.dedent("""
        import json, os
        import torch, torch.distributed as dist
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        mine = torch.full((4,), float(rank), dtype=torch.float64)
        out = torch.empty(4 * world, dtype=torch.float64)
        dist.all_gather_into_tensor(out, mine)
        tmax = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        omp = torch.tensor([float(os.environ.get("OMP_NUM_THREADS", "0"))], dtype=torch.float64)
        dist.all_reduce(omp, op=dist.ReduceOp.MIN)
        if rank == 0:
            print(json.dumps({"world": world, "gathered": out.tolist(), "tmax": float(tmax), "omp_min": float(omp), "local_rank": os.environ["LOCAL_RANK"]}))
        dist.barrier(); dist.destroy_process_group()
    """))
    drv = tmp_path / "drv8.py"
    drv.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {str(ROOT)!r})
        import bench
        raise SystemExit(bench.spawn_ranks(8, {str(stub)!r}, [], timeout_s=280))
    """))
    env = {k: v for k, v in CLEAN.items() if k != "OMP_NUM_THREADS"}
    r = subprocess.run([sys.executable, str(drv)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["world"] == 8 and line["tmax"] == 8.0 and line["local_rank"] == "0"
    assert line["gathered"] == [float(r_) for r_ in range(8) for _ in range(4)]
    assert line["omp_min"] == float(max(1, (os.cpu_count() or 1) // 8))


def test_spawn_ranks_returns_the_childs_failure(tmp_path):
    stub = tmp_path / "fail_stub.py"
    stub.write_text("import sys; sys.exit(7)\n")
    sys.path.insert(0, str(ROOT))
    import bench
    rc = bench.spawn_ranks(2, str(stub), [], timeout_s=240)
    assert rc != 0


def test_plain_invocation_without_enough_gpus_is_one_json_error_line():
    """`python bench.py --gpus 2` on a host with < 2 GPUs (this container has none): refuse before any GPU call."""
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("host has the GPUs")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=CLEAN,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "error" in line and line["n_gpus"] == 2 and line["visible_gpus"] == torch.cuda.device_count()

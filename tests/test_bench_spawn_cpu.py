"""bench.py's self-spawn path (`python bench.py --gpus N` without a launcher re-executes itself under
torch.distributed.run, the launch line of scripts/grpo_pickscore.sh:7-11): the command it builds starts N ranks that
rendezvous on 127.0.0.1 and see each other.  World size 2 over gloo on the CPU; the stub stands in for the script body."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_command_starts_n_ranks_that_see_each_other(tmp_path):
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(textwrap.dedent("""
        import argparse, json, os
        import torch, torch.distributed as dist
        ap = argparse.ArgumentParser(); ap.add_argument("--gpus", type=int); ap.add_argument("--steps", type=int)
        a = ap.parse_args()
        world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
        assert world == a.gpus and os.environ["MASTER_ADDR"] == "127.0.0.1" and "LOCAL_RANK" in os.environ
        dist.init_process_group("gloo")
        ones = torch.ones(1); dist.all_reduce(ones)
        if rank == 0:
            print(json.dumps({"ranks_seen": int(ones.item()), "steps": a.steps}))
        dist.destroy_process_group()
    """))
    cmd = bench.spawn_command(["--gpus", "2", "--steps", "5"], 2, script=str(stub))
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line) == {"ranks_seen": 2, "steps": 5}


def test_gpus_gt_1_without_launcher_takes_the_spawn_path(monkeypatch):
    """No WORLD_SIZE in the environment + --gpus 2 -> self_spawn (which refuses politely here: no GPUs), never the old assert."""
    import pytest
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "GPU(s) visible" in str(e.value)

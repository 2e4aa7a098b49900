"""The self-launcher behind `bench.py --gpus N`: N ranks through torch.distributed.run on 127.0.0.1, started before the
parent touches a GPU.  Exercised on the CPU at world size 2 with a gloo script."""
import os
import subprocess
import sys

from cuda_raytracing_amd import launch
from helpers import ROOT

SCRIPT = '''
import os, sys
import torch
import torch.distributed as dist
dist.init_process_group("gloo")
t = torch.tensor([dist.get_rank() + 1.0])
dist.all_reduce(t)
if dist.get_rank() == 0:
    print("RESULT", dist.get_world_size(), int(t.item()), sys.argv[1:], flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(int(os.environ.get("FAIL_RANK", "-1")) == int(os.environ["RANK"]))
'''


def test_rank_command_shape():
    cmd = launch.rank_command("bench.py", 4, ["--gpus", "4", "--steps", "2"], port=29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5:] == [os.path.abspath("bench.py"), "--gpus", "4", "--steps", "2"]


def test_two_ranks_start_and_their_status_comes_back(tmp_path):
    script = tmp_path / "ranks.py"
    script.write_text(SCRIPT)
    driver = (f"import sys; sys.path.insert(0, {ROOT!r}); from cuda_raytracing_amd.launch import spawn_ranks; "
              f"sys.exit(spawn_ranks({str(script)!r}, 2, ['--gpus', '2']))")
    p = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "RESULT 2 3 ['--gpus', '2']" in p.stdout
    env = dict(os.environ, FAIL_RANK="1")
    p = subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0


def test_bench_hands_over_to_the_launcher_before_importing_torch(tmp_path, monkeypatch):
    """`bench.py --gpus 2` with no WORLD_SIZE must call spawn_ranks first: the parent never initialises a GPU."""
    seen = {}
    monkeypatch.setattr(launch, "spawn_ranks", lambda script, n, argv, env=None: seen.update(script=script, n=n, argv=list(argv)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    sys.modules.pop("bench", None)
    sys.path.insert(0, ROOT)
    import bench
    before = set(sys.modules)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen["n"] == 2 and seen["argv"] == ["--gpus", "2", "--steps", "1"] and seen["script"].endswith("bench.py")
    assert "cuda_raytracing_amd.device" not in set(sys.modules) - before

"""Start one process per GPU for a script that was called with --gpus N and no torch.distributed environment.

The ranks are started through `python -m torch.distributed.run` as a child of the calling process, which must not have
touched a GPU yet (it only waits and hands the child's exit status on): a process that has initialised the GPU must
never be replaced or forked into ranks.  Rendezvous is on 127.0.0.1 with a port that was free a moment ago.
"""
import os
import socket
import subprocess
import sys


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_command(script, n_ranks, argv, port=None):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(script)] + list(argv)


def spawn_ranks(script, n_ranks, argv, env=None):
    """Run `script argv...` as n_ranks ranks; returns the launcher's exit status.  Output goes where this process's goes."""
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    e.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(rank_command(script, n_ranks, argv), env=e).returncode

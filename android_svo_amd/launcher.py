"""`bench.py --gpus N` / `bench_c4.py --gpus N` started plainly (no RANK / WORLD_SIZE in the environment): the parent
process starts the N ranks itself and relays rank 0's JSON line.

This module imports neither torch nor the HIP binding: the parent never touches the GPU (no HIP call, no
`torch.cuda.is_available()`), it only starts `python -m torch.distributed.run --nproc-per-node N <script> <args>` as a
CHILD process (never `os.exec*`), waits for it and exits with its status."""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys


def launched_by_torchrun() -> bool:
    return "RANK" in os.environ or "WORLD_SIZE" in os.environ or "LOCAL_RANK" in os.environ


def free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _end_group(proc, grace_s: float = 5.0) -> str:
    """SIGTERM to the process group `proc` leads (torch.distributed.run tears its workers down), SIGKILL to what is left after
    grace_s; returns what the children had written.  Only ever the group this module created (start_new_session)."""
    try:
        pgid = os.getpgid(proc.pid)
    except ProcessLookupError:
        pgid = None
    out = ""
    if pgid is not None and pgid == proc.pid:
        try:
            os.killpg(pgid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        try:
            out, _ = proc.communicate(timeout=grace_s)
            return out or ""
        except subprocess.TimeoutExpired:
            try:
                os.killpg(pgid, signal.SIGKILL)
            except ProcessLookupError:
                pass
    else:
        proc.kill()
    try:
        out, _ = proc.communicate(timeout=grace_s)
    except subprocess.TimeoutExpired:
        out = ""
    return out or ""


def self_launch(script: str, argv, n_ranks: int, timeout_s: float | None = None) -> int:
    """Start n_ranks ranks of `script argv` under torch.distributed.run on this node (rendezvous on 127.0.0.1), pass the
    children's stdout (rank 0's one JSON line) and stderr through, return their exit status."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script] + list(argv)
    # its own session = its own process group: on a timeout the group this process started is ended as a whole (torchrun's
    # workers inherit the stdout pipe; killing torchrun alone would orphan them on their GPUs and block the read below)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=None, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        out = _end_group(proc)
        sys.stderr.write("launcher: %d ranks of %s did not finish in %s s\n" % (n_ranks, os.path.basename(script), timeout_s))
        sys.stdout.write(out or "")
        return 124
    json_lines = [l for l in (out or "").splitlines() if l.startswith("{")]
    other = [l for l in (out or "").splitlines() if l and not l.startswith("{")]
    if other:
        sys.stderr.write("\n".join(other) + "\n")
    for l in json_lines:
        sys.stdout.write(l + "\n")
    sys.stdout.flush()
    return proc.returncode

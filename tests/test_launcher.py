"""`bench.py --gpus N` started plainly starts its own ranks (android_svo_amd/launcher.py).  CPU-side check of the launcher:
N ranks under torch.distributed.run on 127.0.0.1, rank 0's one JSON line relayed, the children's status returned, and the
parent never imports torch or the HIP binding (it must not touch the GPU before it starts children)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = textwrap.dedent('''
    import json, os, sys
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    import torch
    t = torch.tensor([float(os.environ["RANK"]) + 1.0])
    dist.all_reduce(t)
    print("noise from rank", os.environ["RANK"])
    if int(os.environ["RANK"]) == 0:
        print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"]), "sum": float(t.item()), "args": sys.argv[1:]}))
    dist.destroy_process_group()
    sys.exit(int(sys.argv[1]) if os.environ["RANK"] == "1" else 0)
''')

PARENT = textwrap.dedent('''
    import sys
    sys.path.insert(0, %r)
    from android_svo_amd import launcher
    assert not launcher.launched_by_torchrun()
    rc = launcher.self_launch(sys.argv[1], sys.argv[2:], 2, timeout_s=200)
    assert "torch" not in sys.modules and "android_svo_amd.hip" not in sys.modules, "the launching process must stay off the GPU stack"
    sys.exit(rc)
''') % ROOT


def _run(tmp_path, status):
    script = tmp_path / "rank_script.py"
    script.write_text(RANK_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", PARENT, str(script), str(status), "--flag"], env=env, capture_output=True, text=True, timeout=280)


def test_self_launch_relays_one_json_line_and_the_status(tmp_path):
    p = _run(tmp_path, 0)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1, p.stdout                     # only rank 0's JSON line reaches stdout; the rest goes to stderr
    d = json.loads(lines[0])
    assert d == {"n_gpus": 2, "sum": 3.0, "args": ["0", "--flag"]}
    assert "noise from rank" in p.stderr


def test_self_launch_returns_a_failing_rank_status(tmp_path):
    p = _run(tmp_path, 3)
    assert p.returncode != 0


HANG_SCRIPT = textwrap.dedent('''
    import os, sys, time
    open(os.path.join(sys.argv[1], "pid_%s" % os.environ["RANK"]), "w").write(str(os.getpid()))
    time.sleep(600)
''')

TIMEOUT_PARENT = textwrap.dedent('''
    import sys
    sys.path.insert(0, %r)
    from android_svo_amd import launcher
    sys.exit(launcher.self_launch(sys.argv[1], sys.argv[2:], 2, timeout_s=20))
''') % ROOT


def test_ranks_that_hang_are_ended_as_a_group_on_timeout(tmp_path):
    """A timeout ends torch.distributed.run AND its workers (own process group: SIGTERM, then SIGKILL): no rank is left behind
    holding a GPU or the stdout pipe, and the launcher returns 124 promptly."""
    import time
    script = tmp_path / "hang.py"
    script.write_text(HANG_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", TIMEOUT_PARENT, str(script), str(tmp_path)], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 124, (p.returncode, p.stderr[-1500:])
    assert time.time() - t0 < 60
    pids = [int((tmp_path / ("pid_%d" % r)).read_text()) for r in range(2) if (tmp_path / ("pid_%d" % r)).exists()]
    assert len(pids) == 2, "the ranks did not start within the timeout"
    time.sleep(1.0)
    for pid in pids:                                     # the exact processes this test's launcher started
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, "rank process %d outlived the launcher's timeout" % pid

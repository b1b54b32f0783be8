"""`python3 bench.py --gpus N` starts its own ranks: the launcher's plumbing without a GPU - the children's
rank environment, the relayed line, the exit code - and the timing plumbing (barrier / max / gather over gloo) with two
CPU processes started by that launcher's own rendezvous."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, **env):
    e = dict(os.environ, **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=120)


def test_launcher_starts_one_child_per_gpu_with_the_rendezvous_in_its_environment():
    r = run(["--gpus", "2", "--steps", "3", "--launch-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and [k["rank"] for k in d["ranks"]] == [0, 1]
    for k in d["ranks"]:
        assert k["local_rank"] == k["rank"] and k["world_size"] == 2
        assert k["master_addr"] == "127.0.0.1" and k["master_port"] == d["master_port"]


def test_launcher_also_wraps_a_single_rank():
    r = run(["--launcher", "--launch-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout)
    assert d["n_gpus"] == 1 and d["ranks"] == [dict(rank=0, local_rank=0, world_size=1, master_addr="127.0.0.1", master_port=d["master_port"])]


def test_launcher_exits_non_zero_when_a_rank_does():
    r = run(["--gpus", "3", "--launch-only"], OFDG_BENCH_TEST_FAIL_RANK="1")
    assert r.returncode == 5, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with 5" in r.stderr


def test_launcher_drains_its_children_while_they_run():
    """A rank that prints more than a pipe holds (64 KB) must not block on the launcher: its output is read while it runs."""
    r = run(["--gpus", "2", "--launch-only"], OFDG_BENCH_TEST_PAD="300000")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert [k["rank"] for k in d["ranks"]] == [0, 1]


def test_launcher_starts_again_when_the_port_was_taken():
    """The rendezvous port is probed by the launcher and bound by rank 0 later: when somebody takes it in between, rank 0 exits
    with bench.EXIT_PORT_TAKEN and the launcher repeats the whole start on another port."""
    r = run(["--gpus", "2", "--launch-only"], OFDG_BENCH_TEST_PORT_TAKEN="1")
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    assert "starting again on another one" in r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert [k["rank"] for k in d["ranks"]] == [0, 1]


def test_the_parent_does_not_touch_the_gpu_stack():
    """The launcher must not import torch (let alone torch.cuda) or load libofdg.so: its children initialise the GPU."""
    code = ("import sys, runpy\n"
            "sys.argv = ['bench.py', '--gpus', '2', '--launch-only']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nexcept SystemExit as e:\n    assert not e.code, e.code\n"
            "bad = [m for m in sys.modules if m == 'torch' or m.startswith('torch.') or 'optical-flow' in m]\n"
            "assert not bad, bad\nprint('PARENT_CLEAN')\n" % BENCH)
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "PARENT_CLEAN" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


WORKER = r'''
import os, sys, json
sys.path.insert(0, os.environ["OFDG_ROOT"])
import bench
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
pl = bench.Plumbing(world)
assert "gloo" in pl.how, pl.how
pl.barrier()
assert pl.reduce(10.0 + rank, "max") == 10.0 + world - 1
assert pl.reduce(10.0 + rank, "min") == 10.0
assert pl.gather_ints([rank * 32, 64 + rank * 32]) == [[r * 32, 64 + r * 32] for r in range(world)]
pl.store().set("k%d" % rank, b"x")
pl.close()
if rank == 0:
    print("PLUMBING_OK")
'''


def test_timing_plumbing_over_gloo_two_ranks(tmp_path):
    import socket
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, OFDG_ROOT=ROOT, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    assert "PLUMBING_OK" in outs[0][0]

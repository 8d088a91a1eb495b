"""bench.py --gpus N without a launcher around it (the driver's invocation): the parent stays GPU-free, starts N ranks
with the rendezvous in their environment, relays their output and returns the worst exit code.  CPU only: the ranks
here are stand-in children that print their environment."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHILD = ("import os, sys, json\n"
         "r = int(os.environ['RANK'])\n"
         "print(json.dumps({k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT',"
         " 'ARROWSPACE_BENCH_SHARED_GPU')} | {'argv': sys.argv[1:]}), flush=True)\n"
         "sys.exit(5 if r == 1 and '--fail' in sys.argv else 0)\n")


def _run(n, ndev, extra=()):
    code = ("import sys, bench\n"
            "sys.exit(bench.launch(%d, ['--gpus', '%d', '--steps', '3'] + %r, child=[sys.executable, '-c', %r]))\n" % (n, n, list(extra), CHILD))
    env = dict(os.environ, ARROWSPACE_BENCH_NDEV=str(ndev), PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    return p.returncode, [json.loads(line) for line in p.stdout.splitlines() if line.startswith("{")]


def test_parent_never_imports_torch_or_the_library():
    """The launcher runs before `import torch` / `import pyarrowspace_amd` in main(): a process that has initialised the
    GPU must not start (let alone become) another program."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("sys.exit(launch(") < main.index("import torch") < main.index("import pyarrowspace_amd")
    head = src[: src.index("def main():")]
    launch_src = head[head.index("def launch("): head.index("PARITY = {")]
    assert "import torch" not in launch_src and "pyarrowspace_amd import" not in launch_src


def test_one_rank_per_device():
    rc, lines = _run(4, 8)
    assert rc == 0 and len(lines) == 4
    assert sorted(l["RANK"] for l in lines) == ["0", "1", "2", "3"]
    assert all(l["LOCAL_RANK"] == l["RANK"] and l["WORLD_SIZE"] == "4" and l["MASTER_ADDR"] == "127.0.0.1" for l in lines)
    assert len({l["MASTER_PORT"] for l in lines}) == 1 and all(l["ARROWSPACE_BENCH_SHARED_GPU"] is None for l in lines)
    assert all(l["argv"] == ["--gpus", "4", "--steps", "3"] for l in lines)


def test_rehearsal_on_fewer_devices_and_worst_exit_code():
    rc, lines = _run(3, 1, ["--fail"])
    assert rc == 5                                              # the worst rank's code
    assert sorted(l["RANK"] for l in lines) == ["0", "1", "2"]
    assert all(l["LOCAL_RANK"] == "0" and l["ARROWSPACE_BENCH_SHARED_GPU"] == "1" for l in lines)


def test_visible_devices_follows_the_environment(monkeypatch):
    import bench
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,3")
    assert bench.visible_devices() == 2
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    assert bench.visible_devices() >= 0

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


def test_live_traffic_passes_with_a_stand_in_profiler(monkeypatch, tmp_path):
    """bench.live_traffic(): one child per counter under `rocprofv3 --pmc <C> ... -- python bench.py --traffic-probe ...`,
    summarised per kernel with the gfx950 correction (2 x FETCH_SIZE KiB + WRITE_SIZE KiB).  The profiler here is a
    stand-in script that writes the counter CSV rocprofv3 would (no GPU on this box)."""
    import stat

    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text("""#!%s
import csv, os, sys
a = sys.argv[1:]
c, d, child = a[a.index("--pmc") + 1], a[a.index("-d") + 1], a[a.index("--") + 1:]
assert "--traffic-probe" in child and "--no-cpu-baseline" in child and child[child.index("--n") + 1] == "5000"
os.makedirs(os.path.join(d, "host"), exist_ok=True)
with open(os.path.join(d, "host", "1_counter_collection.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Counter_Name", "Kernel_Name", "Counter_Value"])
    for _ in range(4):
        w.writerow([c, "void as::scan_dma_kernel<6, 4, true>(as::ScanArgs)", 1000.0 if c == "FETCH_SIZE" else 10.0])
    w.writerow([c, "void as::scan_gemm_kernel<1>(as::GemmArgs)", 3000.0 if c == "FETCH_SIZE" else 500.0])
    w.writerow([c, "void other::kernel()", 7.0])
""" % sys.executable)
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    for k in [k for k in os.environ if k.startswith("ROCPROF")]:
        monkeypatch.delenv(k)
    got, note = bench.live_traffic(["--n", "5000", "--d", "64"], timeout_s=60)
    assert got == {"scan_dma_kernel": (2 * 1000 + 10) * 1024, "scan_gemm_kernel": (2 * 3000 + 500) * 1024}, (got, note)
    assert note.startswith("live:")
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")          # already under a profiler: no nesting
    assert bench.live_traffic([], timeout_s=5) == (None, "already under a profiler")


def test_live_traffic_reports_a_failing_pass(monkeypatch, tmp_path):
    import stat

    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\necho no device >&2\nexit 3\n")
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    for k in [k for k in os.environ if k.startswith("ROCPROF")]:
        monkeypatch.delenv(k)
    assert bench.live_traffic([], timeout_s=30) == (None, "FETCH_SIZE pass exited with 3")


def test_live_traffic_ends_a_pass_that_runs_out_of_time(monkeypatch, tmp_path):
    """The stand-in profiler starts a grandchild and sleeps: after the timeout both are gone (one process group)."""
    import stat
    import time

    import bench
    pidfile = tmp_path / "grandchild.pid"
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\nsleep 60 &\necho $! > %s\nsleep 60\n" % pidfile)
    fake.chmod(fake.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    for k in [k for k in os.environ if k.startswith("ROCPROF")]:
        monkeypatch.delenv(k)
    t0 = time.time()
    assert bench.live_traffic([], timeout_s=2) == (None, "FETCH_SIZE pass timed out after 2 s")
    assert time.time() - t0 < 20
    pid = int(pidfile.read_text())
    for _ in range(50):
        try:
            os.kill(pid, 0)
        except OSError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("the pass's grandchild survived the timeout")


HANG_CHILD = ("import os, sys, time\n"
              "r = int(os.environ['RANK'])\n"
              "open(os.path.join(sys.argv[-1], 'pid%d' % r), 'w').write(str(os.getpid()))\n"
              "if r == 1:\n"
              "    time.sleep(0.5); sys.exit(7)\n"
              "time.sleep(600)\n")   # the peers of a dead rank: stuck in a rendezvous / collective it never enters


def test_a_failed_rank_takes_the_others_down(tmp_path):
    """One rank dies (import error, out of memory, RCCL init): the launcher returns ITS code within seconds instead of
    waiting on ranks blocked in a collective, and no rank outlives it (each rank is its own process group)."""
    import time
    code = ("import sys, bench\n"
            "sys.exit(bench.launch(3, [%r], child=[sys.executable, '-c', %r]))\n" % (str(tmp_path), HANG_CHILD))
    env = dict(os.environ, ARROWSPACE_BENCH_NDEV="1", PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None)
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode == 7, (p.returncode, p.stderr[-500:])
    assert time.monotonic() - t0 < 60
    for r in range(3):
        pid = int(open(tmp_path / ("pid%d" % r)).read())
        gone = False
        for _ in range(100):
            try:
                os.kill(pid, 0)
            except ProcessLookupError:
                gone = True
                break
            time.sleep(0.05)
        assert gone, "rank %d (pid %d) survived the launcher" % (r, pid)


def test_launch_time_limit(tmp_path):
    code = ("import sys, bench\n"
            "sys.exit(bench.launch(2, [%r], child=[sys.executable, '-c', 'import time; time.sleep(600)']))\n" % str(tmp_path))
    env = dict(os.environ, ARROWSPACE_BENCH_NDEV="1", PYTHONPATH=ROOT, ARROWSPACE_BENCH_LAUNCH_TIMEOUT="1")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode == 124 and "still running" in p.stderr

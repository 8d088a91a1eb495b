#!/usr/bin/env python3
"""Re-entrant single-query search from T host threads on ONE space (gang scans: callers that arrive together share a pass over
the items): queries/s per T, the members-per-scan counters, and a check that every thread got the serial answers.
    python tools/thread_bench.py [N] [D] [per-thread queries]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pyarrowspace_amd as asp


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    dev = torch.device("cuda:0")
    X = bench.make_data(n, d, 42, dev)
    Q = bench.make_queries(X, 512, 43)
    gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
    want = [aspace.search(q, gl, 0.62) for q in Q]
    for nthr in (1, 2, 3, 4, 1):
        bad = []
        c0 = aspace.gang_counters()

        def worker(t):
            for i in range(per):
                j = (t * 131 + i) % len(Q)
                if aspace.search(Q[j], gl, 0.62) != want[j]:
                    bad.append((t, i))

        ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthr)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.perf_counter() - t0
        c1 = aspace.gang_counters()
        print(f"threads={nthr}: {nthr * per / dt:.0f} q/s, scans by members {[b - a for a, b in zip(c0, c1)]}, wrong answers {len(bad)}", flush=True)
    native(aspace, gl, Q, want, per)


_DRIVER = None


def driver():
    """tools/probe/libthread_driver.so (built on first use with g++ against the library's header)."""
    global _DRIVER
    if _DRIVER is None:
        import ctypes as C
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        so = os.path.join(root, "tools", "probe", "libthread_driver.so")
        src = os.path.join(root, "tools", "probe", "thread_driver.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I" + os.path.join(root, "include"), src, "-o", so,
                                   "-L" + os.path.join(root, "pyarrowspace_amd"), "-larrowspace_hip", "-pthread",
                                   "-Wl,-rpath," + os.path.join(root, "pyarrowspace_amd")])
        L = C.CDLL(so)
        L.as_thread_driver.restype = C.c_double
        L.as_thread_driver.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        _DRIVER = L
    return _DRIVER


def native_rate(aspace, gl, Q, tau, nthr, per, first=None):
    """nthr native threads x per searches over the queries Q (thread t starts at query 131 t) -> (queries/s, errors, scans by members).
    first: expected index of every query's best hit (checked by the driver), or None."""
    import ctypes as C
    L = driver()
    Qc = np.ascontiguousarray(Q, dtype=np.float64)
    topk = min(int(gl.graph_params["topk"]), aspace.nitems)
    c0 = aspace.gang_counters()
    err = C.c_int64(0)
    fp = first.ctypes.data if first is not None else None
    dt = L.as_thread_driver(aspace._h, gl._h, Qc.ctypes.data, len(Qc), Qc.shape[1], float(tau), int(nthr), int(per), topk, fp, C.byref(err))
    c1 = aspace.gang_counters()
    return nthr * per / dt, int(err.value), [b - a for a, b in zip(c0, c1)]


def native(aspace, gl, Q, want, per):
    """The same closed loops from native threads (tools/probe/thread_driver.cpp: no interpreter lock between the calls)."""
    first = np.array([w[0][0] for w in want], dtype=np.int64)
    for nthr in (1, 2, 3, 4, 1):
        rate, errs, gangs = native_rate(aspace, gl, Q, 0.62, nthr, per, first)
        print(f"native threads={nthr}: {rate:.0f} q/s, scans by members {gangs}, wrong first hits {errs}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Single-query search on the unfriendly distributions of bench.py (isotropic rows, hierarchical clusters on a common
direction, x100-scaled rows under the cosine distance with eps = 10) at the headline shape: queries/s, fallbacks by cause, the
scan's operand and duration, and a brute-force check of 32 of the timed queries.  One JSON line per distribution.

    python tools/dist_sweep.py [--n 1000000 --d 768 --steps 200 --warmup 20 --tau 0.62] [--kinds isotropic hier scaled100]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--k", type=int, default=25)
    ap.add_argument("--topk", type=int, default=15)
    ap.add_argument("--tau", type=float, nargs="*", default=[0.62])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--metric", default="l2")
    ap.add_argument("--kernel", default="gaussian")
    ap.add_argument("--kinds", nargs="*", default=["isotropic", "hier", "scaled100"])
    args = ap.parse_args()
    import torch

    import bench

    device = torch.device("cuda", 0)
    for kind in args.kinds:
        for tau in args.tau:
            a = argparse.Namespace(**vars(args))
            a.tau = tau
            out = bench.run_distribution(kind, args.n, args.d, a, device)
            print(json.dumps(dict(kind=kind, tau=tau, n=args.n, d=args.d, **out)), flush=True)


if __name__ == "__main__":
    main()

#!/bin/bash
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
STALL_PROBE_TAG=default python tools/stall_probe.py 2>/dev/null | grep variant
STALL_PROBE_TAG=torch-collectives ARROWSPACE_PY_COLLECTIVES=1 python tools/stall_probe.py 2>/dev/null | grep variant
STALL_PROBE_TAG=gc-freeze STALL_PROBE_GC=freeze python tools/stall_probe.py 2>/dev/null | grep variant
STALL_PROBE_TAG=gc-disabled STALL_PROBE_GC=disable python tools/stall_probe.py 2>/dev/null | grep variant
STALL_PROBE_TAG=gc-freeze-no-warmup ARROWSPACE_NO_COMM_WARMUP=1 STALL_PROBE_GC=freeze python tools/stall_probe.py 2>/dev/null | grep variant

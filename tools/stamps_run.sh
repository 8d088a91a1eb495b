#!/bin/bash
# Runs on the GPU box: rebuilds the library with in-kernel stamps, prints the phase times of the finish kernels, rebuilds without.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
make -s clean >/dev/null 2>&1; make -s STAMPS=1 >/dev/null 2>&1 || { echo "stamps build failed"; exit 1; }
python tools/stamps.py "$@" 2>&1 | grep -v amdgpu.ids
make -s clean >/dev/null 2>&1; make -s >/dev/null 2>&1

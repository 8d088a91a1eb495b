#!/bin/bash
# what bounds the tile scan: (a) steady state at 4x the rows, (b) the scorer-candidate work switched off (ABLATION build, wrong results)
cd "$(dirname "$0")/.."
timeout -k 10 400 python tools/tile_geom.py 4000000 768 208 4216 2>&1 | grep geom
make -s clean >/dev/null 2>&1; make -s -j 16 ABLATION=1 > /dev/null 2>&1 || { echo "ablation build failed"; exit 1; }
for dbg in 0 1 2 4 7; do
  echo "ARROWSPACE_SC_DBG=$dbg"
  ARROWSPACE_SC_DBG=$dbg timeout -k 10 300 python tools/tile_geom.py 1000000 768 208 2>&1 | grep geom | tail -1
done

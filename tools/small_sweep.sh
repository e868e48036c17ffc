#!/bin/bash
# Run ON THE GPU BOX: the reference's default frame (400x225, 100 spp, 255 spheres) under several job sizes, with
# maxDepth 50 and 8 (no long paths), then configs[1].
cd "$GRAFT_REPO_ROOT"
for depth in 50 8; do for mc in 4 2 1; do
  echo "DEPTH=$depth PT_MIN_CHUNK=$mc"
  PT_MIN_CHUNK=$mc DEPTH=$depth W=400 H=225 SPHERES=250 SPP=100 REPS=30 timeout -k 10 60 python3 tools/perf_run.py | tail -1 | sed 's/.*\]//' || exit 1
done; done
echo configs1
REPS=6 timeout -k 10 120 python3 tools/perf_run.py || exit 1

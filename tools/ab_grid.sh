#!/bin/bash
# Run ON THE GPU BOX: the grid search (ACCEL=1) on configs[1] and configs[4] with several builds of the library.
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do for lib in "$@"; do
  echo "== $lib configs1 grid"
  ACCEL=1 PT_HIP_LIB=$PWD/tools/build/$lib REPS=5 timeout -k 10 120 python3 tools/perf_run.py | tail -1 | sed 's/kernel ms.*\]//' || exit 1
  echo "== $lib configs4 grid"
  ACCEL=1 SPHERES=10000 EXTENT=45 SPP=256 PT_HIP_LIB=$PWD/tools/build/$lib REPS=3 timeout -k 10 120 python3 tools/perf_run.py | tail -1 | sed 's/kernel ms.*\]//' || exit 1
done; done

#!/bin/bash
# Run ON THE GPU BOX: the same frames with several builds of the library (PT_HIP_LIB), on one box.
# Usage: tools/ab.sh libA.so libB.so ...   (files under tools/build/)
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do for lib in "$@"; do for mc in ${CHUNKS:-4}; do
  echo "== $lib small PT_MIN_CHUNK=$mc"
  PT_MIN_CHUNK=$mc PT_HIP_LIB=$PWD/tools/build/$lib W=400 H=225 SPHERES=250 SPP=100 REPS=30 timeout -k 10 60 python3 tools/perf_run.py | tail -1 | sed 's/.*\]//' || exit 1
  done
  if [ -z "$SMALL_ONLY" ]; then
  echo "== $lib configs1"
  PT_HIP_LIB=$PWD/tools/build/$lib REPS=5 timeout -k 10 120 python3 tools/perf_run.py | sed 's/kernel ms.*\]//' || exit 1
  echo "== $lib configs3"
  PT_HIP_LIB=$PWD/tools/build/$lib DIFFUSE=0.1 METAL=0.5 DEPTH=100 REPS=5 timeout -k 10 120 python3 tools/perf_run.py | tail -1 | sed 's/kernel ms.*\]//' || exit 1
  fi
  if [ -n "$GRID" ]; then
  echo "== $lib configs1 grid"
  ACCEL=1 PT_HIP_LIB=$PWD/tools/build/$lib REPS=5 timeout -k 10 120 python3 tools/perf_run.py | tail -2 | sed 's/kernel ms.*\]//' || exit 1
  fi
done; done

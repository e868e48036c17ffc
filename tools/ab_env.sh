#!/bin/bash
# Run ON THE GPU BOX: one environment variable swept over values, for several builds; small frame and configs[1].
# Usage: VAR=PT_TAIL_ROUNDS VALUES="1 2 4" [LIBS="libA.so libB.so"] tools/ab_env.sh
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do for lib in ${LIBS:-default}; do for v in $VALUES; do
  echo "== $lib $VAR=$v"
  [ "$lib" != default ] && export PT_HIP_LIB=$PWD/tools/build/$lib
  export $VAR=$v
  W=400 H=225 SPHERES=250 SPP=100 REPS=30 timeout -k 10 60 python3 tools/perf_run.py 2>&1 | tail -1 | sed 's/.*\]//' || exit 1
  if [ -z "$SMALL_ONLY" ]; then
  REPS=5 timeout -k 10 120 python3 tools/perf_run.py 2>&1 | grep -v "prof\]\|^hash" | sed 's/kernel ms.*\]//' || exit 1
  fi
done; done; done

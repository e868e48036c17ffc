#!/bin/bash
# Run ON THE GPU BOX: the small jobs at the end of a launch (PT_FINE_CHUNK samples per pixel, PT_FINE_SHARE percent of one
# large job per wave; PT_TARGET_JOBS sets the size of the large ones), small frame and configs[1], for several builds.
# Usage: [FINE="0 2"] [SHARE="100"] [TARGET="262144"] [LIBS="libA.so libB.so"] tools/ab_fine.sh
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do for lib in ${LIBS:-default}; do for tj in ${TARGET:-262144}; do for fc in ${FINE:-0 2}; do for fs in ${SHARE:-100}; do
  echo "== $lib target $tj fine chunk $fc share $fs"
  [ "$lib" != default ] && export PT_HIP_LIB=$PWD/tools/build/$lib
  export PT_FINE_CHUNK=$fc PT_FINE_SHARE=$fs PT_TARGET_JOBS=$tj
  W=400 H=225 SPHERES=250 SPP=100 REPS=30 timeout -k 10 60 python3 tools/perf_run.py 2>&1 | tail -1 | sed 's/.*\]//' || exit 1
  if [ -z "$SMALL_ONLY" ]; then
  REPS=5 timeout -k 10 120 python3 tools/perf_run.py 2>&1 | grep -v "prof\]" | sed 's/kernel ms.*\]//' || exit 1
  fi
done; done; done; done; done

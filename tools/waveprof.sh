#!/bin/bash
# Run ON THE GPU BOX: per-wave timing of the end of a launch (make waveprof first), the 400x225 frame and configs[1].
cd "$GRAFT_REPO_ROOT"
export PT_HIP_LIB=$PWD/tools/build/libpt_hip_waveprof.so
echo "# 400x225, 100 spp, 255 spheres"
W=400 H=225 SPHERES=250 SPP=100 REPS=3 timeout -k 10 60 python3 tools/perf_run.py 2>&1 | grep "prof\]" | tail -9 || exit 1
echo "# configs[1]: 1920x1080, 100 spp, 505 spheres"
REPS=3 timeout -k 10 60 python3 tools/perf_run.py 2>&1 | grep "prof\]" | tail -9 || exit 1
echo "# configs[1] with the end-of-launch measures off: PT_FINE_CHUNK=0 (wave priorities stay)"
PT_FINE_CHUNK=0 REPS=3 timeout -k 10 60 python3 tools/perf_run.py 2>&1 | grep "prof\]" | tail -9

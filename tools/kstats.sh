#!/bin/bash
# Run ON THE GPU BOX: per-kernel durations (rocprofv3 --kernel-trace --stats) of tools/perf_run.py under the current environment.
# Usage: [W=.. H=.. SPP=.. ...] tools/kstats.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1
rm -rf gpurun_out/${tag}_ks
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_ks -- python3 tools/perf_run.py > gpurun_out/${tag}_ks.log 2>&1 || exit 1
grep -h "k_render\|k_tail" gpurun_out/${tag}_ks/*/*kernel_stats.csv | cut -d, -f1,2,4,6,7

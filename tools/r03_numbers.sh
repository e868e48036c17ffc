#!/bin/bash
# Run ON THE GPU BOX: the round's numbers for DESIGN.md section 6 (one box, one call).
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 bench.py --steps 10 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || exit 1
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --workload configs2 --no-extras --no-cpu-baseline > gpurun_out/r03_bench_configs2.json 2>> gpurun_out/r03_bench.err || exit 1
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --workload configs4 --no-extras --no-cpu-baseline > gpurun_out/r03_bench_configs4.json 2>> gpurun_out/r03_bench.err || exit 1
timeout -k 10 300 python3 bench.py --gpus 2 --steps 5 --warmup 2 --rehearse-on-one-gpu --no-extras --no-cpu-baseline > gpurun_out/r03_bench_2ranks_rehearsal.json 2>> gpurun_out/r03_bench.err || exit 1
timeout -k 10 120 path_tracer_amd/lib/pathtracer tools/configs/4k_p3.txt /tmp/o3.ppm && timeout -k 10 120 path_tracer_amd/lib/pathtracer tools/configs/4k_p6.txt /tmp/o6.ppm

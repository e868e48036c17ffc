#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats and counter passes of the default bench (the full scan,
# k_render<0> + k_tail), and of the grid search (k_render<2>) on configs[1] and configs[4].
# Each counter set gets a run of its own (rocprofv3 does not split counters over passes); the program comes directly
# after `--` (no env / sh -c hop).
# Usage: tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_stats, gpurun_out/<tag>_pmc{1,2,3,4}, gpurun_out/<tag>_g{1,4}_*
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
PMC1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
PMC2="SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_F32 SQ_WAVES_EQ_64"
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_pmc* gpurun_out/${tag}_g*
# (the kernel-time pass runs more launches: the first launch of a process is 8 % slower than the others and would weigh a quarter)
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras > gpurun_out/${tag}_stats.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc $PMC1 --output-format csv -d gpurun_out/${tag}_pmc1 -- $B > gpurun_out/${tag}_pmc1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc $PMC2 --output-format csv -d gpurun_out/${tag}_pmc2 -- $B > gpurun_out/${tag}_pmc2.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc3 -- $B > gpurun_out/${tag}_pmc3.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc4 -- $B > gpurun_out/${tag}_pmc4.log 2>&1
echo "scan collected"
# the grid search: configs[1] (g1) and configs[4] (g4: 10,000 spheres on a half extent of 45, 256 spp)
export ACCEL=1 REPS=3
for g in g1 g4; do
  if [ $g = g4 ]; then export SPHERES=10000 EXTENT=45 SPP=256; fi
  G="python3 tools/perf_run.py"
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${g}_stats -- $G > gpurun_out/${tag}_${g}_stats.log 2>&1
  timeout -s KILL 150 rocprofv3 --pmc $PMC1 --output-format csv -d gpurun_out/${tag}_${g}_pmc1 -- $G > gpurun_out/${tag}_${g}_pmc1.log 2>&1
  timeout -s KILL 150 rocprofv3 --pmc $PMC2 --output-format csv -d gpurun_out/${tag}_${g}_pmc2 -- $G > gpurun_out/${tag}_${g}_pmc2.log 2>&1
  timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/${tag}_${g}_pmc3 -- $G > gpurun_out/${tag}_${g}_pmc3.log 2>&1
  timeout -s KILL 150 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d gpurun_out/${tag}_${g}_pmc4 -- $G > gpurun_out/${tag}_${g}_pmc4.log 2>&1 || echo "pass ${g}_pmc4 failed (counter names?)"
done
echo collected

#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel stats and counter passes of the default bench.
# Each counter set gets a run of its own (rocprofv3 does not split counters over passes).
# Usage: tools/collect_profiles.sh <tag>     -> gpurun_out/<tag>_stats, gpurun_out/<tag>_pmc{1,2,3,4}
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_pmc*
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- $B > gpurun_out/${tag}_stats.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${tag}_pmc1 -- $B > gpurun_out/${tag}_pmc1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_VALU_MFMA_F32 SQ_WAVES_EQ_64 --output-format csv -d gpurun_out/${tag}_pmc2 -- $B > gpurun_out/${tag}_pmc2.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc3 -- $B > gpurun_out/${tag}_pmc3.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc4 -- $B > gpurun_out/${tag}_pmc4.log 2>&1
echo collected

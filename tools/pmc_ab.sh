#!/bin/bash
# Run ON THE GPU BOX: vector-instruction counters of k_render<0> on configs[1] for several builds of the library
# (files under tools/build/), one rocprofv3 --pmc pass each; prints the largest dispatch of each (the 1080p launch).
# Usage: tools/pmc_ab.sh libA.so libB.so ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  d=gpurun_out/pmcab_${lib%.so}
  rm -rf $d
  PT_HIP_LIB=$PWD/tools/build/$lib REPS=1 timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_LDS --output-format csv -d $d -- python3 tools/perf_run.py > $d.log 2>&1 || { echo "pass failed for $lib"; tail -5 $d.log; exit 1; }
  python3 - $d $lib <<'PY'
import csv, glob, sys, collections
d, lib = sys.argv[1], sys.argv[2]
best = collections.defaultdict(float)
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k_render" in row["Kernel_Name"]:
            k = row["Counter_Name"]; best[k] = max(best[k], float(row["Counter_Value"]))
dur = 0.0
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        if "k_render" in row["Kernel_Name"]:
            dur = max(dur, (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e6)
print(lib, "ms=%.3f clock_GHz=%.3f" % (dur, best.get("GRBM_GUI_ACTIVE", 0) / 8 / (dur * 1e-3) / 1e9 if dur else 0), " ".join("%s=%.4e" % (k, v) for k, v in sorted(best.items())))
PY
done

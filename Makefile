# Builds everything in-tree: the gfx950 library, the host-side C++ library, the oracle.
HIPCC ?= hipcc
ARCH ?= gfx950
LIBDIR := path_tracer_amd/lib
# no atomic optimizer: every atomic of these kernels is issued by one lane or to per-lane addresses; the wave-aggregated
# form hipcc substitutes only adds vector instructions (measured: -0.3 % kernel time without it)
# -amdgpu-sched-strategy=max-ilp: the scheduler interleaves the two independent chains of seven packed FMAs of a quad (the
# default strategy keeps each chain together and pays a wait state, an s_nop, between dependent packed FMAs: twelve per quad
# in the scan loop); measured 1.5 % on configs[1], 1.6 % on configs[3], 2.5 % on the 400x225 frame, same frames
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-sched-strategy=max-ilp

all: $(LIBDIR)/libpt_hip.so check host oracle

$(LIBDIR)/libpt_hip.so: path_tracer_amd/csrc/pt_hip.hip path_tracer_amd/csrc/pt_device.h path_tracer_amd/csrc/pt_grid.h include/pt_hip.h Makefile
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $< -ldl

host:
	@if [ -f path_tracer_amd/host/Makefile ]; then $(MAKE) -C path_tracer_amd/host; fi

oracle:
	$(MAKE) -C oracle

SRC := path_tracer_amd/csrc/pt_hip.hip path_tracer_amd/csrc/pt_device.h path_tracer_amd/csrc/pt_grid.h include/pt_hip.h

asm: tools/build/pt_hip.s
tools/build/pt_hip.s: $(SRC) Makefile
	@mkdir -p tools/build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -o $@ $< 2> tools/build/pt_hip.resources.txt
tools/build/pt_hip_stats.s: $(SRC) Makefile
	@mkdir -p tools/build
	$(HIPCC) $(HIPFLAGS) -DPT_STATS -S --cuda-device-only -o $@ $< 2> /dev/null

# A required part of the build: the hand-written scalar loads of pt_device.h are only correct if hipcc leaves their
# destination registers alone until the wait (tools/check_sload.py follows every path through the compiled kernels, for
# the default flags and for the PT_STATS diagnostic build), and the sample-loop kernels must not spill
# (tools/check_resources.py). A compiler upgrade has to pass both.
check: tools/build/.checked
tools/build/.checked: tools/build/pt_hip.s tools/build/pt_hip_stats.s tools/check_sload.py tools/check_resources.py
	python3 tools/check_sload.py tools/build/pt_hip.s
	python3 tools/check_sload.py tools/build/pt_hip_stats.s
	python3 tools/check_resources.py tools/build/pt_hip.resources.txt
	@touch $@

# the diagnostic build with wave-level counters (tools/stats_run.py: PT_HIP_LIB=tools/build/libpt_hip_stats.so)
stats: $(SRC)
	@mkdir -p tools/build
	$(HIPCC) $(HIPFLAGS) -DPT_STATS -shared -o tools/build/libpt_hip_stats.so $< -ldl

# the diagnostic build with per-wave timestamps (when each wave of k_render ran out of jobs and left, how long the groups
# of k_tail took): PT_HIP_LIB=tools/build/libpt_hip_waveprof.so python3 tools/perf_run.py prints one line per launch
waveprof: $(SRC)
	@mkdir -p tools/build
	$(HIPCC) $(HIPFLAGS) -DPT_WAVEPROF -shared -o tools/build/libpt_hip_waveprof.so $< -ldl

clean:
	rm -rf $(LIBDIR)/*.so oracle/_build tools/build

.PHONY: all host oracle asm check stats waveprof clean

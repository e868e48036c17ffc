# Builds everything in-tree: the gfx950 library, the host-side C++ library, the oracle.
HIPCC ?= hipcc
ARCH ?= gfx950
LIBDIR := path_tracer_amd/lib
# no atomic optimizer: every atomic of these kernels is issued by one lane or to per-lane addresses; the wave-aggregated
# form hipcc substitutes only adds vector instructions (measured: -0.3 % kernel time without it)
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None

all: $(LIBDIR)/libpt_hip.so host oracle

$(LIBDIR)/libpt_hip.so: path_tracer_amd/csrc/pt_hip.hip path_tracer_amd/csrc/pt_device.h path_tracer_amd/csrc/pt_grid.h include/pt_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $< -ldl

host:
	@if [ -f path_tracer_amd/host/Makefile ]; then $(MAKE) -C path_tracer_amd/host; fi

oracle:
	$(MAKE) -C oracle

asm: path_tracer_amd/csrc/pt_hip.hip path_tracer_amd/csrc/pt_device.h path_tracer_amd/csrc/pt_grid.h
	@mkdir -p tools/build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -o tools/build/pt_hip.s $< 2> tools/build/pt_hip.resources.txt

clean:
	rm -rf $(LIBDIR)/*.so oracle/_build tools/build

.PHONY: all host oracle asm clean

# Top-level build: the product library + drop-in CLIs (gfx950 only) and the oracle.
# Host code is C (gcc, the HIP runtime through its C API); hipcc compiles the kernels and links.
#
#   make            -> voice_synth_amd/lib/libvoicesynth.so, voice_synth_amd/bin/{flowgen_shimmer,vowel}
#   make oracle     -> oracle/liboracle.so and, when /root/reference exists, oracle/_ref/*
#
# -ffp-contract=off everywhere: the float/double rounding sequence is part of the parity
# contract; the fused variant of the filter uses explicit fma() calls.

ROCM    ?= /opt/rocm
HIPCC   ?= $(ROCM)/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950

PKG     := voice_synth_amd
CSRC    := $(PKG)/csrc
LIBDIR  := $(PKG)/lib
BINDIR  := $(PKG)/bin

KERNEL_HDRS := $(CSRC)/vs_device.h $(CSRC)/vs_dev_primitives.h $(CSRC)/vs_dev_generator.h $(CSRC)/vs_dev_filter.h include/voice_synth.h
HIPFLAGS := -O3 --offload-arch=$(ARCH) -ffp-contract=off -fPIC -std=c++17 -Wall -Wno-unused-function
CFLAGS   := -O2 -ffp-contract=off -fno-fast-math -fPIC -Wall -Wextra -Wno-unused-parameter
HOSTFLAGS := -std=gnu11 $(CFLAGS) -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include
HOST_HDRS := $(CSRC)/vs_device.h $(CSRC)/vs_internal.h $(CSRC)/vs_planhost.h include/voice_synth.h

LIB := $(LIBDIR)/libvoicesynth.so

all: $(LIB) clis

$(LIBDIR) $(BINDIR):
	mkdir -p $@

$(CSRC)/vs_host.o: $(CSRC)/vs_host.c $(CSRC)/vs_tables.h include/voice_synth.h
	$(CC) $(CFLAGS) -c -o $@ $<

$(CSRC)/vs_kernels.o: $(CSRC)/vs_kernels.hip $(KERNEL_HDRS)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

# the one-wave kernel once more with 16 utterances per wavefront: periods beyond the 64-column ring
$(CSRC)/vs_kernels_narrow.o: $(CSRC)/vs_kernels.hip $(KERNEL_HDRS)
	$(HIPCC) $(HIPFLAGS) -DVS_GROUP_LANES=16 -c -o $@ $<

# the host side of the library: plain C against the HIP runtime's C API
$(CSRC)/vs_api.o: $(CSRC)/vs_api.c $(HOST_HDRS)
	$(CC) $(HOSTFLAGS) -c -o $@ $<

$(CSRC)/vs_planhost.o: $(CSRC)/vs_planhost.c $(CSRC)/vs_planhost.h $(CSRC)/vs_device.h include/voice_synth.h
	$(CC) -std=gnu11 $(CFLAGS) -c -o $@ $<

$(CSRC)/vs_delivery.o: $(CSRC)/vs_delivery.c $(HOST_HDRS)
	$(CC) $(HOSTFLAGS) -c -o $@ $<

$(CSRC)/vs_node.o: $(CSRC)/vs_node.c $(CSRC)/vs_commguard.h $(HOST_HDRS)
	$(CC) $(HOSTFLAGS) -c -o $@ $<

$(CSRC)/vs_commguard.o: $(CSRC)/vs_commguard.c $(CSRC)/vs_commguard.h
	$(CC) -std=gnu11 $(CFLAGS) -c -o $@ $<

$(LIB): $(CSRC)/vs_host.o $(CSRC)/vs_planhost.o $(CSRC)/vs_kernels.o $(CSRC)/vs_kernels_narrow.o $(CSRC)/vs_api.o $(CSRC)/vs_delivery.o $(CSRC)/vs_node.o $(CSRC)/vs_commguard.o | $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lm -lpthread -ldl

clis: $(BINDIR)/flowgen_shimmer $(BINDIR)/vowel $(BINDIR)/vs_batch $(BINDIR)/vs_bench

$(BINDIR)/%: $(PKG)/cli/%.c $(PKG)/cli/cli_common.h $(LIB) | $(BINDIR)
	$(CC) -O2 -ffp-contract=off -Wall -Iinclude -o $@ $< -L$(LIBDIR) -lvoicesynth -lm -lpthread -Wl,-rpath,'$$ORIGIN/../lib'

oracle:
	$(MAKE) -C oracle all

# resource usage (VGPR/SGPR/LDS/occupancy) of every kernel
resources:
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c -o /dev/null $(CSRC)/vs_kernels.hip

clean:
	rm -f $(CSRC)/*.o $(LIB) $(LIBDIR)/libvoicesynth_*.so $(BINDIR)/flowgen_shimmer $(BINDIR)/vowel $(BINDIR)/vs_batch $(BINDIR)/vs_bench
	$(MAKE) -C oracle clean

.PHONY: all clis oracle resources clean diag

# diagnostic build with s_memtime stamps (never shipped, never timed): tools/diag_bench.py
diag: $(LIBDIR)/libvoicesynth_diag.so
$(CSRC)/vs_kernels_diag.o: $(CSRC)/vs_kernels.hip $(KERNEL_HDRS)
	$(HIPCC) $(HIPFLAGS) -DVS_DIAG -c -o $@ $<
$(LIBDIR)/libvoicesynth_diag.so: $(CSRC)/vs_kernels_diag.o $(CSRC)/vs_kernels_narrow.o $(CSRC)/vs_api.o $(CSRC)/vs_delivery.o $(CSRC)/vs_node.o $(CSRC)/vs_commguard.o $(CSRC)/vs_host.o $(CSRC)/vs_planhost.o | $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -lm -lpthread -ldl

# A/B variants of the library for same-box comparisons (tools/gpu_ab.sh):
#   make variant NAME=sleep2 DEFS="-DVS_POLL_SLEEP=2"   ->  lib/libvoicesynth_sleep2.so   (select with VS_LIB)
variant: $(CSRC)/vs_kernels_narrow.o $(CSRC)/vs_api.o $(CSRC)/vs_delivery.o $(CSRC)/vs_node.o $(CSRC)/vs_commguard.o $(CSRC)/vs_host.o $(CSRC)/vs_planhost.o | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) $(DEFS) -c -o $(CSRC)/vs_kernels_$(NAME).o $(CSRC)/vs_kernels.hip
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $(LIBDIR)/libvoicesynth_$(NAME).so $(CSRC)/vs_kernels_$(NAME).o $(CSRC)/vs_kernels_narrow.o $(CSRC)/vs_api.o $(CSRC)/vs_delivery.o $(CSRC)/vs_node.o $(CSRC)/vs_commguard.o $(CSRC)/vs_host.o $(CSRC)/vs_planhost.o -lm -lpthread -ldl

# device listing of the shipped kernels (same flags) for tools/isa_loops.py
isa:
	mkdir -p build && $(HIPCC) $(HIPFLAGS) -Iinclude -I$(CSRC) -S --cuda-device-only -o build/vs_kernels.s $(CSRC)/vs_kernels.hip

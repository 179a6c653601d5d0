# Builds the product library (HIP, gfx950) and the test oracle.
#   make            libdoppler_hip.so + oracle
#   make lib        doppler_amd/lib/libdoppler_hip.so only
#   make oracle     oracle/liboracle.so, oracle/check_sincosf, oracle/_ref (if reference mounted)
ROCM    ?= /opt/rocm
HIPCC   ?= $(ROCM)/bin/hipcc
ARCH    ?= gfx950
CSRC    := doppler_amd/csrc
LIBDIR  := doppler_amd/lib
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
            -Wall -Wno-unused-function -Iinclude

LIB := $(LIBDIR)/libdoppler_hip.so
API  := dpx_context dpx_resident dpx_operators dpx_plans dpx_stream
OBJS := $(LIBDIR)/dpx_kernels.o $(API:%=$(LIBDIR)/%.o) $(LIBDIR)/dpx_planner.o $(LIBDIR)/dpx_simulate.o $(LIBDIR)/orbit.o $(LIBDIR)/schedule.o

all: lib cli oracle cpptest

lib: $(LIB)

# -amdgpu-kernarg-preload-count=16: the first 16 dwords of kernel arguments arrive in SGPRs,
# so the one-shot wavefronts issue their loads without an s_load round trip (see rows_kernel).
$(LIBDIR)/dpx_kernels.o: $(CSRC)/dpx_kernels.hip $(CSRC)/dpx_sincos.h $(CSRC)/dpx_types.h include/doppler_hip.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -mllvm -amdgpu-kernarg-preload-count=16 -c $< -o $@

# the C ABI: host C++ over the HIP runtime API (no device code)
$(API:%=$(LIBDIR)/%.o): $(LIBDIR)/%.o: $(CSRC)/%.cpp $(CSRC)/dpx_internal.h $(CSRC)/dpx_planner.h $(CSRC)/dpx_types.h $(CSRC)/host/orbit.h $(CSRC)/host/schedule.h include/doppler_hip.h include/doppler_hip_debug.h include/doppler_hip_host.h
	@mkdir -p $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIBDIR)/dpx_planner.o $(LIBDIR)/dpx_simulate.o: $(LIBDIR)/%.o: $(CSRC)/%.cpp $(CSRC)/dpx_planner.h $(CSRC)/dpx_types.h
	@mkdir -p $(LIBDIR)
	g++ -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -c $< -o $@

$(LIBDIR)/%.o: $(CSRC)/host/%.cpp $(CSRC)/host/orbit.h $(CSRC)/host/schedule.h
	@mkdir -p $(LIBDIR)
	g++ -O2 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -Wl,-rpath,$(ROCM)/lib -Wl,-soname,libdoppler_hip.so

# the `doppler` command (stdin -> GPU -> stdout), linked against the C ABI library
BINDIR := doppler_amd/bin
cli: $(BINDIR)/doppler

$(BINDIR)/doppler: $(CSRC)/cli/main.cpp $(CSRC)/cli/args.cpp $(CSRC)/cli/args.h $(CSRC)/host/orbit.h $(CSRC)/host/schedule.h include/doppler_hip.h include/doppler_hip_debug.h include/doppler_hip_host.h $(LIB)
	@mkdir -p $(BINDIR)
	g++ -O2 -std=c++17 -ffp-contract=off -Wall -pthread -Iinclude $(CSRC)/cli/main.cpp $(CSRC)/cli/args.cpp \
	    -o $@ -L$(LIBDIR) -ldoppler_hip -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,$(ROCM)/lib

# the reference's in-file tests against the C++ mirror of doppler::dsp (needs the oracle as checker)
tests/cpp/test_dsp: tests/cpp/test_dsp.cpp include/doppler_dsp.hpp include/doppler_hip.h $(LIB) oracle
	g++ -O2 -std=c++17 -ffp-contract=off -Wall -o $@ tests/cpp/test_dsp.cpp -L$(LIBDIR) -ldoppler_hip -Loracle -loracle \
	    -Wl,-rpath,'$$ORIGIN/../../doppler_amd/lib' -Wl,-rpath,'$$ORIGIN/../../oracle' -Wl,-rpath,$(ROCM)/lib -lm

cpptest: tests/cpp/test_dsp

# host-only fuzz of the planner under AddressSanitizer + UBSan (no GPU, no HIP)
tests/cpp/test_planner_fuzz: tests/cpp/test_planner_fuzz.cpp $(CSRC)/dpx_planner.cpp $(CSRC)/dpx_simulate.cpp $(CSRC)/dpx_planner.h $(CSRC)/dpx_types.h
	g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -fno-fast-math -Wall \
	    -o $@ tests/cpp/test_planner_fuzz.cpp $(CSRC)/dpx_planner.cpp $(CSRC)/dpx_simulate.cpp

planner-fuzz: tests/cpp/test_planner_fuzz
	tests/cpp/test_planner_fuzz 300

# the closed-form first reset against the candidate-by-candidate scan (host only)
tests/cpp/test_find_reset: tests/cpp/test_find_reset.cpp $(CSRC)/dpx_planner.cpp $(CSRC)/dpx_simulate.cpp $(CSRC)/dpx_planner.h $(CSRC)/dpx_types.h
	g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -Wall -o $@ tests/cpp/test_find_reset.cpp $(CSRC)/dpx_planner.cpp $(CSRC)/dpx_simulate.cpp

find-reset: tests/cpp/test_find_reset
	tests/cpp/test_find_reset 1

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf $(LIBDIR) $(BINDIR)
	$(MAKE) -C oracle clean

.PHONY: all lib cli oracle cpptest planner-fuzz find-reset clean

# Builds libagz.so (HIP, gfx950) and the CPU oracle (test infrastructure).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CS := agogo_amd/csrc
OUT := agogo_amd/lib
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-uninitialized -Iinclude
# engine.hip holds the bit-exact MCTS arithmetic: no fused multiply-add contraction (Go/amd64 never fuses)
ENGINE_FLAGS := -ffp-contract=off

SRCS := $(CS)/ctx.hip $(CS)/net.hip $(CS)/engine.hip $(CS)/train.hip $(CS)/examples.hip $(CS)/comm.hip
OBJS := $(OUT)/ctx.o $(OUT)/net.o $(OUT)/engine.o $(OUT)/train.o $(OUT)/examples.o $(OUT)/comm.o
HDRS := $(wildcard $(CS)/*.hpp) include/agz.h

all: $(OUT)/libagz.so oracle tests/cpp/az_learn_ttt tests/cpp/gtp_main tests/fake_rccl/librccl_fake.so

# TEST INFRASTRUCTURE: a process-per-rank stand-in for librccl (AGZ_RCCL_LIB) so the N > 1 exchange executes on a one-GPU box
tests/fake_rccl/librccl_fake.so: tests/fake_rccl/fake_rccl.cpp
	g++ -std=c++17 -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $< -o $@ -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib

# host-side C++ mirror of the reference API (agogo_amd/host/agogo.hpp) exercised over the C ABI
tests/cpp/az_learn_ttt: tests/cpp/az_learn_ttt.cpp agogo_amd/host/agogo.hpp include/agz.h $(OUT)/libagz.so
	g++ -std=c++17 -O1 -Iinclude $< -o $@ -L$(OUT) -lagz -Wl,-rpath,'$$ORIGIN/../../agogo_amd/lib'

# GTP front end over the C ABI (agogo_amd/host/gtp.hpp)
tests/cpp/gtp_main: tests/cpp/gtp_main.cpp agogo_amd/host/gtp.hpp agogo_amd/host/agogo.hpp include/agz.h $(OUT)/libagz.so
	g++ -std=c++17 -O1 -Iinclude $< -o $@ -L$(OUT) -lagz -Wl,-rpath,'$$ORIGIN/../../agogo_amd/lib'

$(OUT)/ctx.o: $(CS)/ctx.hip $(HDRS)
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OUT)/net.o: $(CS)/net.hip $(HDRS)
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OUT)/train.o: $(CS)/train.hip $(HDRS)
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OUT)/examples.o: $(CS)/examples.hip $(HDRS)
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OUT)/comm.o: $(CS)/comm.hip $(HDRS)
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(OUT)/engine.o: $(CS)/engine.hip $(HDRS)
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) $(ENGINE_FLAGS) -c $< -o $@
$(OUT)/libagz.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(OUT)/*.o $(OUT)/libagz.so
	$(MAKE) -C oracle clean
.PHONY: all oracle clean

# Build recipe for the MI355X-native alignment engine.
#   make lib     -> vg_amd/libvgamd.so        HIP kernels + C ABI (gfx950)
#   make host    -> vg_amd/libvgamd_host.so   C++ host shim mirroring vg's Aligner interface
#   make oracle  -> oracle/libvgoracle.so     CPU oracle (test infrastructure only)
# Built artefacts are git-ignored; they travel to the GPU box with gpurun.

HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
CC       ?= gcc
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Iinclude -Wall -Wno-unused-function -Wno-unused-value
CXXFLAGS ?= -O2 -std=c++17 -fPIC -Iinclude -Wall
CFLAGS   ?= -O3 -march=x86-64-v2 -std=c11 -fPIC -fopenmp -Iinclude -Wall

LIB_SRCS    := $(wildcard vg_amd/csrc/*.hip) $(wildcard vg_amd/csrc/*.cpp)
LIB_HDRS    := $(wildcard vg_amd/csrc/*.h vg_amd/csrc/*.hpp include/*.h)
HOST_SRCS   := $(wildcard vg_amd/host/*.cpp) $(wildcard vg_amd/host/vg_standin/*.cpp)
HOST_HDRS   := $(wildcard vg_amd/host/*.hpp vg_amd/host/vg_standin/*.hpp include/*.h)
ORACLE_SRCS := $(wildcard oracle/*.c)

all: lib host oracle
lib: vg_amd/libvgamd.so
host: vg_amd/libvgamd_host.so
oracle: oracle/libvgoracle.so

# the C-ABI/packing layer is plain C++; only the .hip files carry device code
LIB_CPP_OBJS := $(patsubst %.cpp,%.o,$(wildcard vg_amd/csrc/*.cpp))
LIB_HIP_OBJS := $(patsubst %.hip,%.o,$(wildcard vg_amd/csrc/*.hip))
vg_amd/csrc/%.o: vg_amd/csrc/%.cpp $(LIB_HDRS)
	$(CXX) $(CXXFLAGS) -O3 -c $< -o $@
vg_amd/csrc/%.o: vg_amd/csrc/%.hip $(LIB_HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
vg_amd/libvgamd.so: $(LIB_CPP_OBJS) $(LIB_HIP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(LIB_CPP_OBJS) $(LIB_HIP_OBJS)

vg_amd/libvgamd_host.so: $(HOST_SRCS) $(HOST_HDRS)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRCS) -ldl

oracle/libvgoracle.so: $(ORACLE_SRCS) include/vgk.h oracle/vgo_haplo.h
	$(CC) $(CFLAGS) -shared -o $@ $(ORACLE_SRCS)

clean:
	rm -f vg_amd/csrc/*.o vg_amd/libvgamd.so vg_amd/libvgamd_host.so oracle/libvgoracle.so

.PHONY: all lib host oracle clean

# test-only: CPU lock-step emulation of the HIP lane code behind the same C ABI
emu: tests/emu/libvgamd_emu.so
EMU_OBJS := $(patsubst vg_amd/csrc/%.cpp,tests/emu/obj/%.o,$(wildcard vg_amd/csrc/*.cpp)) tests/emu/obj/backend_emu.o
tests/emu/obj/%.o: vg_amd/csrc/%.cpp $(LIB_HDRS)
	@mkdir -p tests/emu/obj
	$(CXX) -O2 -g -std=c++17 -fPIC -Iinclude -Wall -c $< -o $@
tests/emu/obj/backend_emu.o: tests/emu/backend_emu.cpp $(LIB_HDRS)
	@mkdir -p tests/emu/obj
	$(CXX) -O2 -g -std=c++17 -fPIC -Iinclude -Wall -c $< -o $@
tests/emu/libvgamd_emu.so: $(EMU_OBJS)
	$(CXX) -shared -o $@ $(EMU_OBJS) -lpthread
.PHONY: emu

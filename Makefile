# Build recipe for the MI355X-native alignment engine.
#   make lib     -> vg_amd/libvgamd.so        HIP kernels + C ABI (gfx950)
#   make host    -> vg_amd/libvgamd_host.so   C++ host shim mirroring vg's Aligner interface
#   make oracle  -> oracle/libvgoracle.so     CPU oracle (test infrastructure only)
# Built artefacts are git-ignored; they travel to the GPU box with gpurun.

HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
CC       ?= gcc
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Iinclude -Wall -Wno-unused-function
CXXFLAGS ?= -O2 -std=c++17 -fPIC -Iinclude -Wall
CFLAGS   ?= -O3 -march=x86-64-v2 -std=c11 -fPIC -fopenmp -Iinclude -Wall

LIB_SRCS    := $(wildcard vg_amd/csrc/*.hip)
LIB_HDRS    := $(wildcard vg_amd/csrc/*.h vg_amd/csrc/*.hpp include/*.h)
HOST_SRCS   := $(wildcard vg_amd/host/*.cpp)
HOST_HDRS   := $(wildcard vg_amd/host/*.hpp include/*.h)
ORACLE_SRCS := $(wildcard oracle/*.c)

all: lib host oracle
lib: vg_amd/libvgamd.so
host: vg_amd/libvgamd_host.so
oracle: oracle/libvgoracle.so

vg_amd/libvgamd.so: $(LIB_SRCS) $(LIB_HDRS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(LIB_SRCS)

vg_amd/libvgamd_host.so: $(HOST_SRCS) $(HOST_HDRS)
	$(CXX) $(CXXFLAGS) -shared -o $@ $(HOST_SRCS) -ldl

oracle/libvgoracle.so: $(ORACLE_SRCS) include/vgk.h
	$(CC) $(CFLAGS) -shared -o $@ $(ORACLE_SRCS)

clean:
	rm -f vg_amd/libvgamd.so vg_amd/libvgamd_host.so oracle/libvgoracle.so

.PHONY: all lib host oracle clean

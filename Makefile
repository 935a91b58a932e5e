# Builds everything in-tree (the .so files travel to the GPU box with the snapshot).
#   make            -> product library + oracle + scene generator
#   make cli        -> pathtracer_amd/hpt_cli             (g++)            headless viewer replacement
#   make lib        -> pathtracer_amd/libhpt.so           (hipcc, gfx950)  the product
#   make oracle     -> oracle/libpt_oracle.so             (gcc)            test infrastructure
#   make scenes     -> scenes/libhpt_scenes.so            (gcc)            test/bench infrastructure
HIPCC ?= /opt/rocm/bin/hipcc
CC    ?= gcc
CXX   ?= g++

# No FMA contraction, IEEE divide/sqrt: the numerics contract of DESIGN.md §Numerics.
# No SLP vectorisation: hipcc packs the triangle test into v_pk ops glued by ~20 v_mov (2 % slower, more VGPRs).
# HPT_SAMPLE_BUDGET_DEFAULT_MB: sample buffer of hpt_render (paths per launch x 12 B; HPT_SAMPLE_BUDGET_MB overrides at run time).
# HPT_NOALLOC_SCRATCH_MB: the static device array hpt_render_noalloc works in.
SAMPLE_BUDGET_MB ?= 2048
NOALLOC_SCRATCH_MB ?= 128
HIPFLAGS = -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off \
           -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -Iinclude -Wall -Wno-unused-function \
           -DHPT_SAMPLE_BUDGET_DEFAULT_MB=$(SAMPLE_BUDGET_MB) -DHPT_NOALLOC_SCRATCH_MB=$(NOALLOC_SCRATCH_MB)
HOSTFLAGS = -O2 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Wall
CFLAGS_ORACLE = -O2 -std=c11 -fPIC -ffp-contract=off -fno-fast-math -fopenmp -Wall

LIB_SRC = pathtracer_amd/csrc/hpt_backend.hip
# kernel A/B builds: make variant EXTRA="-DFOO=1" OUT=build/libhpt_foo.so
EXTRA ?=
OUT ?= build/libhpt_variant.so
HOST_SRC = pathtracer_amd/csrc/host/bvh.cpp pathtracer_amd/csrc/host/obj_reader.cpp \
           pathtracer_amd/csrc/host/image_writer.cpp pathtracer_amd/csrc/host/pathtracer.cpp \
           pathtracer_amd/csrc/host/host_capi.cpp pathtracer_amd/csrc/host/checkpoint.cpp
HOST_OBJ = $(HOST_SRC:.cpp=.o)

all: lib cli oracle scenes fake_rccl

lib: pathtracer_amd/libhpt.so

pathtracer_amd/csrc/host/%.o: pathtracer_amd/csrc/host/%.cpp $(wildcard pathtracer_amd/csrc/host/*.hpp) pathtracer_amd/csrc/hpt_internal.h $(wildcard include/hpt/*.h)
	$(CXX) $(HOSTFLAGS) -c $< -o $@

pathtracer_amd/csrc/hpt_backend.o: $(LIB_SRC) $(wildcard pathtracer_amd/csrc/*.h) $(wildcard include/hpt/*.h)
	$(HIPCC) $(HIPFLAGS) -c $(LIB_SRC) -o $@

# the multi-GPU exchange (RCCL through dlopen: no link-time dependency)
pathtracer_amd/csrc/hpt_comm.o: pathtracer_amd/csrc/hpt_comm.hip pathtracer_amd/csrc/hpt_internal.h $(wildcard include/hpt/*.h)
	$(HIPCC) $(HIPFLAGS) -c pathtracer_amd/csrc/hpt_comm.hip -o $@

pathtracer_amd/libhpt.so: pathtracer_amd/csrc/hpt_backend.o pathtracer_amd/csrc/hpt_comm.o $(HOST_OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC pathtracer_amd/csrc/hpt_backend.o pathtracer_amd/csrc/hpt_comm.o $(HOST_OBJ) -ldl -o $@

variant: $(HOST_OBJ) pathtracer_amd/csrc/hpt_comm.o
	mkdir -p $(dir $(OUT))
	$(HIPCC) $(HIPFLAGS) $(EXTRA) -c $(LIB_SRC) -o $(OUT).o
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(OUT).o pathtracer_amd/csrc/hpt_comm.o $(HOST_OBJ) -ldl -o $(OUT)

cli: pathtracer_amd/hpt_cli
pathtracer_amd/hpt_cli: pathtracer_amd/csrc/host/hpt_cli.cpp pathtracer_amd/libhpt.so
	$(CXX) $(HOSTFLAGS) pathtracer_amd/csrc/host/hpt_cli.cpp -o $@ -Lpathtracer_amd -lhpt -Wl,-rpath,'$$ORIGIN'

oracle: oracle/libpt_oracle.so
oracle/libpt_oracle.so: oracle/pt_oracle.c oracle/pt_oracle.h include/hpt/types.h
	$(CC) $(CFLAGS_ORACLE) -shared oracle/pt_oracle.c -o $@ -lm

scenes: scenes/libhpt_scenes.so
scenes/libhpt_scenes.so: scenes/scenes.c
	$(CC) -O2 -std=c11 -fPIC -ffp-contract=off -shared scenes/scenes.c -o $@ -lm

# test infrastructure: a stand-in for RCCL whose ranks share one process and may share one GPU (tests/test_multi_gpu.py)
fake_rccl: tests/native/libfake_rccl.so
tests/native/libfake_rccl.so: tests/native/fake_rccl.hip
	$(HIPCC) -O2 -std=c++17 --offload-arch=gfx950 -shared -fPIC tests/native/fake_rccl.hip -o $@

# micro-benchmarks behind the roofline ceilings (profiles/): build/gather_exec, build/gather_rate, ...
MICRO = $(patsubst bench/micro/%.hip,build/%,$(wildcard bench/micro/*.hip))
micro: $(MICRO)
build/%: bench/micro/%.hip
	mkdir -p build
	$(HIPCC) -O3 -std=c++17 --offload-arch=gfx950 $< -o $@

clean:
	rm -f pathtracer_amd/libhpt.so oracle/libpt_oracle.so scenes/libhpt_scenes.so $(HOST_OBJ) pathtracer_amd/csrc/hpt_backend.o pathtracer_amd/csrc/hpt_comm.o

.PHONY: all lib cli oracle scenes clean variant micro fake_rccl

# Build of the three shared libraries (no cmake: plain make + hipcc/g++).
#   v-img_amd/lib/libvimg_hip.so   hand-written gfx950 kernels + C ABI   (the product)
#   v-img_amd/lib/libvimg_host.so  host side: scene loading, SAH BVH, post (the product's host)
#   oracle/liboracle.so            CPU restatement of the reference path  (test infrastructure)
ROOT    := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950

LIBDIR  := v-img_amd/lib
HOSTSRC := $(wildcard v-img_amd/host/*.cpp)
HOSTHDR := $(wildcard v-img_amd/host/*.hpp) $(wildcard include/*.h)
HIPSRC  := $(wildcard v-img_amd/csrc/*.hip)
HIPHDR  := $(wildcard v-img_amd/csrc/*.h) $(wildcard include/*.h)
ORASRC  := $(wildcard oracle/*.cpp)
ORAHDR  := $(wildcard oracle/*.h) $(wildcard include/*.h)

# -ffp-contract=off everywhere: +,-,*,/,sqrt must round identically on CPU and GPU; fused
# multiply-adds appear only where the reference writes std::fma.
HOSTFLAGS := -std=c++20 -O2 -fPIC -shared -ffp-contract=off -fopenmp -Wall -Iinclude
ORAFLAGS  := -std=c++20 -O3 -march=x86-64-v3 -fPIC -shared -ffp-contract=off -pthread -Wall -Iinclude
HIPFLAGS  := --offload-arch=$(ARCH) -std=c++20 -O3 -fPIC -shared -ffp-contract=off -fno-slp-vectorize \
             -fno-fast-math -Iinclude -Wall -Wno-unused-function

all: host hip dev oracle oracle-avx2 cli
host: $(LIBDIR)/libvimg_host.so
hip: $(LIBDIR)/libvimg_hip.so
oracle: oracle/liboracle.so
cli: v-img_amd/bin/vimg-amd

$(LIBDIR)/libvimg_host.so: $(HOSTSRC) $(HOSTHDR) Makefile
	@mkdir -p $(LIBDIR)
	$(CXX) $(HOSTFLAGS) $(HOSTSRC) -o $@

# one object per translation unit (make -j compiles them side by side): the ABI unit, one unit per
# render kernel family, the BVH builders
HIPCFLAGS := $(filter-out -shared,$(HIPFLAGS)) -c
OBJDIR    := build/hip
HIPOBJ    := $(patsubst v-img_amd/csrc/%.hip,$(OBJDIR)/%.o,$(HIPSRC))
$(OBJDIR)/%.o: v-img_amd/csrc/%.hip $(HIPHDR) Makefile
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPCFLAGS) $< -o $@
# render_cu_kernel is ONE persistent loop around every stage: MachineLICM hoists the constants of all
# of them (the f64 polynomial coefficients of acos / atan2 / pow ...) in front of the loop, where they
# no longer fit the register file: 533 spilled registers and 700 bytes of scratch per lane with the
# pass, 80 / 72 without (tools/kres.sh)
$(OBJDIR)/k_cu.o $(OBJDIR)/k_cu_early.o $(OBJDIR)/k_cu_diag.o: HIPCFLAGS += -mllvm -disable-machine-licm
$(LIBDIR)/libvimg_hip.so: $(HIPOBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(HIPOBJ) -o $@

# development build: the product's kernels plus the two retired schedulers (round 1's pooled kernel,
# the staged kernel) the GPU tests cross-check against; selected with VIMG_HIP_LIB
DEVOBJDIR := build/hip_dev
DEVOBJ    := $(patsubst v-img_amd/csrc/%.hip,$(DEVOBJDIR)/%.o,$(HIPSRC))
$(DEVOBJDIR)/k_dev.o: v-img_amd/csrc/k_dev.hip $(HIPHDR) Makefile
	@mkdir -p $(DEVOBJDIR)
	$(HIPCC) $(HIPCFLAGS) -DVIMG_DEV_SCHEDULERS=1 $< -o $@
$(DEVOBJDIR)/k_pool4.o: v-img_amd/csrc/k_pool4.hip $(HIPHDR) Makefile
	@mkdir -p $(DEVOBJDIR)
	$(HIPCC) $(HIPCFLAGS) -DVIMG_DEV_SCHEDULERS=1 $< -o $@
$(DEVOBJDIR)/%.o: $(OBJDIR)/%.o
	@mkdir -p $(DEVOBJDIR)
	cp $< $@
dev: v-img_amd/lib/dev/libvimg_hip.so
v-img_amd/lib/dev/libvimg_hip.so: $(DEVOBJ)
	@mkdir -p v-img_amd/lib/dev
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(DEVOBJ) -o $@

# C++ host program (the counterpart of the reference's main): links both libraries by rpath
v-img_amd/bin/vimg-amd: v-img_amd/cli/main.cpp $(LIBDIR)/libvimg_host.so $(LIBDIR)/libvimg_hip.so Makefile
	@mkdir -p v-img_amd/bin
	$(HIPCC) -std=c++20 -O2 -Iinclude v-img_amd/cli/main.cpp -L$(LIBDIR) -lvimg_host -lvimg_hip \
	  -Wl,-rpath,'$$ORIGIN/../lib' -o $@

oracle/liboracle.so: $(ORASRC) $(ORAHDR) Makefile
	$(CXX) $(ORAFLAGS) $(ORASRC) -o $@

# same oracle with the reference's own float libm calls (cosf, acosf ...) instead of the
# double-evaluated forms the GPU can reproduce bit for bit; used by one CPU test.
oracle/liboracle_libmf.so: $(ORASRC) $(ORAHDR) Makefile
	$(CXX) $(ORAFLAGS) -DORACLE_LIBM_FLOAT=1 $(ORASRC) -o $@

# TIMING build of the oracle (never a parity partner): the reference's AVX2 two-sibling slab path
# (include/simd_hit.h:121-156, include/bvh.h:109-116) with the optimisation level and contraction
# default of the reference's release build; x86-64-v3 (AVX2 + FMA) instead of -march=native because
# the library is built in the build container and timed on the GPU box's host CPU.
oracle-avx2: oracle/liboracle_avx2.so
oracle/liboracle_avx2.so: $(ORASRC) $(ORAHDR) Makefile
	$(CXX) -std=c++20 -O3 -march=x86-64-v3 -fPIC -shared -pthread -Wall -Iinclude -DORACLE_AVX2_TIMING=1 $(ORASRC) -o $@

clean:
	rm -rf $(LIBDIR)/*.so $(LIBDIR)/dev oracle/*.so build/hip build/hip_dev

.PHONY: all host hip dev oracle oracle-avx2 cli clean

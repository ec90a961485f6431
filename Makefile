# Builds libtrh.so (HIP kernels + C ABI, gfx950 only) and the oracle's C++ restatement.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := tiny-ram-halo2_amd
CSRC := $(PKG)/csrc
# build id = hash of the library's sources (trh_version() reports it)
BUILD_ID := $(shell cat $(CSRC)/*.hip $(CSRC)/*.h include/trh.h | sha1sum | cut -c1-12)
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function -Wno-unused-result
OBJS := $(CSRC)/capi.o $(CSRC)/msm.o $(CSRC)/ntt.o $(CSRC)/ipa.o $(CSRC)/ipafold.o $(CSRC)/pointfft.o $(CSRC)/domain.o $(CSRC)/scan.o $(CSRC)/expr.o $(CSRC)/lookup.o $(CSRC)/hostio.o $(CSRC)/selftest.o
HDRS := $(CSRC)/field.h $(CSRC)/curve.h $(CSRC)/curve_q4.h $(CSRC)/ctx.h $(CSRC)/hostcombine.h $(CSRC)/copypool.h $(CSRC)/devpool.h $(CSRC)/selftest_kat.h include/trh.h

all: $(PKG)/libtrh.so oracle examples/replay tests/native/multi_ctx_test tests/native/libtrh_q4broken.so

$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# capi.o carries the build id: rebuilt whenever any source of the library changes
$(CSRC)/capi.o: $(CSRC)/capi.hip $(HDRS) $(wildcard $(CSRC)/*.hip)
	$(HIPCC) $(HIPFLAGS) -DTRH_BUILD_ID='"$(BUILD_ID)"' -c $< -o $@

$(PKG)/libtrh.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -ldl -o $@

# The library WITHOUT curve_q4.h's work-around for the ROCm 7.2 DPP-combiner miscompile: trh_init's self-test has to refuse it
# (tests/test_gpu_selftest.py).  Only the two objects that instantiate the quad-lane group law are rebuilt.
$(CSRC)/%.q4b.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -DTRH_TEST_DROP_Q4_WORKAROUND -DTRH_BUILD_ID='"$(BUILD_ID)-q4broken"' -c $< -o $@
tests/native/libtrh_q4broken.so: $(CSRC)/capi.q4b.o $(CSRC)/msm.q4b.o $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(CSRC)/capi.q4b.o $(CSRC)/msm.q4b.o $(filter-out $(CSRC)/capi.o $(CSRC)/msm.o,$(OBJS)) -ldl -o $@

# native (C++17, no Python) driver over include/trh.hpp
examples/replay: examples/replay.cpp include/trh.hpp include/trh.h $(PKG)/libtrh.so
	g++ -O2 -std=c++17 -Wall -Iinclude $< -o $@ -L$(PKG) -ltrh -pthread -Wl,-rpath,'$$ORIGIN/../$(PKG)'

# native test of the context layer (device group, per-thread contexts); run by tests/test_gpu_native.py
tests/native/multi_ctx_test: tests/native/multi_ctx_test.cpp include/trh.h $(PKG)/libtrh.so
	g++ -O1 -std=c++17 -Wall -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include $< -o $@ -L$(PKG) -ltrh -L/opt/rocm/lib -lamdhip64 -pthread \
	    -Wl,-rpath,'$$ORIGIN/../../$(PKG)' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -s -C oracle libtrh_oracle.so

clean:
	rm -f $(OBJS) $(CSRC)/*.q4b.o $(PKG)/libtrh.so tests/native/libtrh_q4broken.so examples/replay tests/native/multi_ctx_test
	$(MAKE) -s -C oracle clean

.PHONY: all oracle clean

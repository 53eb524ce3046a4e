# Builds the product library (HIP, gfx950 only) and the test oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -DGM_FR_MUL_ASM -mllvm -enable-misched=0
SRC := $(wildcard gkr_msm_amd/csrc/*.hip)
OBJ := $(patsubst gkr_msm_amd/csrc/%.hip,build/%.o,$(SRC))
HDR := $(wildcard gkr_msm_amd/csrc/*.inc) $(wildcard gkr_msm_amd/csrc/*.hip.h) $(wildcard gkr_msm_amd/csrc/*.hpp) include/gkrmsm.h
LIB := gkr_msm_amd/libgkrmsm_hip.so

all: $(LIB) oracle examples ubench

$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ) -ldl

build/%.o: gkr_msm_amd/csrc/%.hip $(HDR)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

oracle:
	$(MAKE) -s -C oracle

# plain C callers of the C ABI (no HIP headers, no C++): gcc only
examples: build/examples/pippenger build/examples/gkr_msm_simple build/examples/pippenger_sharded
build/examples/%: examples/%.c include/gkrmsm.h $(LIB)
	@mkdir -p build/examples
	gcc -std=c11 -O2 -Wall -Wextra -pthread -D_POSIX_C_SOURCE=200809L -Iinclude $< -o $@ -Lgkr_msm_amd -lgkrmsm_hip -Wl,-rpath,'$$ORIGIN/../../gkr_msm_amd'

# device-side self-checks that tests/ run on the GPU box (prebuilt here: the 14 x 28 one takes hipcc three minutes); each binary
# carries the digest of its source + the field headers, tests/ubench_util.py rebuilds only when that digest is stale
ubench: build/ubench/fq14_test build/ubench/fr9_mul_test
UBENCH_HDRS := gkr_msm_amd/csrc/fq14.hip.h gkr_msm_amd/csrc/g1.hip.h gkr_msm_amd/csrc/fq.hip.h gkr_msm_amd/csrc/fr9.hip.h gkr_msm_amd/csrc/fr.hip.h \
	gkr_msm_amd/csrc/fq14_mul_gen.inc gkr_msm_amd/csrc/fr9_mul_asm.inc gkr_msm_amd/csrc/fr9_sqr_asm.inc gkr_msm_amd/csrc/fr_mul_asm.inc
build/ubench/%: scripts/ubench/%.hip $(HDR)
	@mkdir -p build/ubench
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -w -mllvm -enable-misched=0 -o $@ $<
	cat $< $(UBENCH_HDRS) | sha256sum | cut -d' ' -f1 > $@.srchash

clean:
	rm -rf build $(LIB)
	$(MAKE) -s -C oracle clean

.PHONY: all oracle examples ubench clean

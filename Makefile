# Builds the gfx950 C-ABI library (in-tree, so it travels with gpurun) and the oracle-side C helpers.
PKG := cross-scale-mae_amd
SRC := $(wildcard $(PKG)/csrc/*.hip)
OBJ := $(patsubst $(PKG)/csrc/%.hip,build/obj/%.o,$(SRC))
LIB := $(PKG)/csmae_hip/libcsmae_hip.so
HIPCC ?= hipcc
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result

all: $(LIB)

# the content hash of csrc/ is baked into the library (csmae_source_hash()): a profile or a bench line can then say which sources the
# kernels it measured were built from.  build/obj/src_hash.txt is rewritten only when the hash moves, so api.o rebuilds exactly then.
SRC_HASH := $(shell python3 tools/csrc_hash.py)
build/obj/src_hash.txt: $(SRC) $(PKG)/csrc/common.h $(PKG)/csrc/gemm_common.h
	@mkdir -p build/obj
	@echo '$(SRC_HASH)' | cmp -s - $@ || echo '$(SRC_HASH)' > $@

build/obj/api.o: $(PKG)/csrc/api.hip $(PKG)/csrc/common.h build/obj/src_hash.txt
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -DCSMAE_SRC_HASH='"$(SRC_HASH)"' -c $< -o $@

build/obj/%.o: $(PKG)/csrc/%.hip $(PKG)/csrc/common.h $(PKG)/csrc/gemm_common.h
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC $(OBJ) -o $@

probe: tools/probe_gfx950.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=gfx950 -O2 $< -o build/probe_gfx950

clean:
	rm -rf build/obj $(LIB)
.PHONY: all clean probe

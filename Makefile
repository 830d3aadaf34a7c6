# Builds the gfx950 C-ABI library (in-tree, so it travels with gpurun) and the oracle-side C helpers.
PKG := cross-scale-mae_amd
SRC := $(wildcard $(PKG)/csrc/*.hip)
OBJ := $(patsubst $(PKG)/csrc/%.hip,build/obj/%.o,$(SRC))
LIB := $(PKG)/csmae_hip/libcsmae_hip.so
HIPCC ?= hipcc
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result

all: $(LIB)

build/obj/%.o: $(PKG)/csrc/%.hip $(PKG)/csrc/common.h
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

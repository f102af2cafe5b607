# Builds libunivst_hip.so (gfx950 only) and the oracle helpers.  `python __graft_entry__.py build` calls this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := univst_amd/csrc
OBJ   := build/obj
LIB   := univst_amd/lib/libunivst_hip.so
SRCS  := $(wildcard $(CSRC)/*.hip)
OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(SRCS))
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude

all: $(LIB)

$(OBJ)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.h) include/univst.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p univst_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# AddressSanitizer run of the host side of the C ABI: the library is rebuilt with -fsanitize=address on the HOST
# compilation only (device code is unaffected) into build/asan/, and tests/test_abi.py (symbol table, error strings, argument
# validation — no GPU needed) runs against it with libasan preloaded.  `make asan`.
ASAN_LIB := build/asan/libunivst_hip_asan.so
ASAN_RT  := $(shell $(HIPCC) --offload-arch=$(ARCH) -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null)
$(ASAN_LIB): $(SRCS) $(wildcard $(CSRC)/*.h) include/univst.h
	@mkdir -p build/asan
	$(HIPCC) $(FLAGS) -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -shared -shared-libsan -o $@ $(SRCS)
asan: $(ASAN_LIB)
	LD_PRELOAD=$(ASAN_RT) ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 UNIVST_LIB=$(abspath $(ASAN_LIB)) python -m pytest tests/test_abi.py -x -q

clean:
	rm -rf build $(LIB)
.PHONY: all clean asan

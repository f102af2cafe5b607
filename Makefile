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

clean:
	rm -rf build $(LIB)
.PHONY: all clean

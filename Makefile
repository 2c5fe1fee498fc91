# Builds the C-ABI shared library of hand-written gfx950 kernels (in-tree; travels to the GPU box).
HIPCC ?= hipcc
ARCH ?= gfx950
CSRC := stonkgs_amd/csrc
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(SRCS:.hip=.o)
LIB := $(CSRC)/libstonk_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Iinclude -I$(CSRC) -Wno-unused-result -ffp-contract=fast

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/gemm_common.h $(CSRC)/stonk_flags.h include/stonk_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -f $(OBJS) $(LIB)
.PHONY: all clean

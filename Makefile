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

# Host-side AddressSanitizer build of the launchers (CPU only: the device code is compiled as usual, the host code -
# argument validation, launch geometry, descriptor handling - is instrumented). Load it with the sanitizer runtime
# preloaded: tests/test_host_cpu.py::test_asan_host_build_of_the_launchers does.
ASAN_DIR := build/asan
ASAN_OBJS := $(patsubst $(CSRC)/%.hip,$(ASAN_DIR)/%.o,$(SRCS))
ASAN_LIB := $(ASAN_DIR)/libstonk_hip_asan.so
ASANFLAGS := --offload-arch=$(ARCH) -O1 -g -fPIC -std=c++17 -Iinclude -I$(CSRC) -Wno-unused-result -ffp-contract=fast \
             -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer

$(ASAN_DIR)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/gemm_common.h $(CSRC)/stonk_flags.h include/stonk_hip.h
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(ASANFLAGS) -c $< -o $@

$(ASAN_LIB): $(ASAN_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Xarch_host -fsanitize=address -shared-libsan -o $@ $(ASAN_OBJS)

asan: $(ASAN_LIB)

clean:
	rm -f $(OBJS) $(LIB)
	rm -rf $(ASAN_DIR)
.PHONY: all clean asan

# Builds libmimo_hip.so (gfx950) and the oracle's C helpers.  `python -c "import __graft_entry__ as g; g.build()"` calls this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := mimo_unet_amd/csrc
SRCS := $(CSRC)/conv3x3.hip $(CSRC)/conv_bf16x3.hip $(CSRC)/conv_wide.hip $(CSRC)/wgrad_split.hip $(CSRC)/elementwise.hip $(CSRC)/optim.hip $(CSRC)/plan.hip $(CSRC)/ops_api.hip
OBJS := $(SRCS:.hip=.o)
LIB := mimo_unet_amd/libmimo_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function

all: $(LIB)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/sched.h $(CSRC)/elementwise.h include/mimo_hip.h
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean

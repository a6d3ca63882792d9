# Builds libmimo_hip.so (gfx950).  `python -c "import __graft_entry__ as g; g.build()"` calls this.  (The oracle is pure
# Python on torch's CPU operators and the reference has no native sources: nothing else to compile.)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := mimo_unet_amd/csrc
SRCS := $(CSRC)/conv3x3.hip $(CSRC)/conv_thin.hip $(CSRC)/conv_bf16x3.hip $(CSRC)/conv_wide.hip $(CSRC)/wgrad_split.hip $(CSRC)/elementwise.hip $(CSRC)/optim.hip $(CSRC)/plan.hip $(CSRC)/ops_api.hip
OBJS := $(SRCS:.hip=.o)
LIB := mimo_unet_amd/libmimo_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function

all: $(LIB)

# every compile leaves hipcc's per-kernel resource remarks next to the object (csrc/*.res): scripts/check_resources.py
# fails the build when a kernel that counts its vector-memory operations by hand (LDS-DMA) touches scratch
$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/tile_sched.h $(CSRC)/elementwise.h include/mimo_hip.h
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c $< -o $@ 2> $(@:.o=.res) || (grep -v "remark:" $(@:.o=.res) >&2; false)
	@grep -E "warning:|error:" -A3 $(@:.o=.res) >&2 || true

$(LIB): $(OBJS) $(CSRC)/exports.map
	python3 scripts/check_resources.py $(OBJS:.o=.res)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -Wl,--version-script=$(CSRC)/exports.map -o $@ $(OBJS)

clean:
	rm -f $(OBJS) $(OBJS:.o=.res) $(LIB)

.PHONY: all clean

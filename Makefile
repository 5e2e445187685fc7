# Builds libxlbhip.so (HIP, gfx950) and the C oracle.  No cmake: hipcc/gcc directly.
HIPCC     ?= /opt/rocm/bin/hipcc
ARCH      ?= gfx950
CSRC      := xlb_amd/csrc
OBJDIR    ?= build/obj
LIB       ?= xlb_amd/lib/libxlbhip.so
# -ffp-contract=off: fp32/fp64 results are bit-identical to the oracle's operation order
EXTRA     ?=
HIPFLAGS  := $(EXTRA) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -Iinclude
SRCS      := api.hip comm.cpp step_d2q9_bgk.hip step_d2q9_kbc.hip step_d3q19_bgk.hip step_d3q27_bgk.hip step_d3q27_kbc.hip step_d3q27_kbc_fast.hip \
             step_d2q9_ext.hip step_d3q19_ext.hip step_d3q27_ext.hip step2_d3q19.hip step2_d3q19_strips.hip step2_d3q27.hip yardstick.hip
OBJS      := $(addprefix $(OBJDIR)/,$(addsuffix .o,$(basename $(SRCS))))
HDRS      := $(wildcard $(CSRC)/*.hpp) include/xlbhip.h

all: $(LIB) oracle

# the two-step kernel packs its fp32 pairs by hand (cell.hpp: bgk_packed_pairs); hipcc's SLP vectorizer on top of that
# scrambles the sequential moment sums into packed adds + moves (measured +2 % kernel time)
$(OBJDIR)/step2_d3q19.o $(OBJDIR)/step2_d3q19_strips.o $(OBJDIR)/step2_d3q27.o: HIPFLAGS += -fno-slp-vectorize

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p xlb_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

oracle: oracle/liblbmref.so

oracle/liblbmref.so: oracle/lbm_ref.c oracle/lbm_ref_body.inc
	gcc -O2 -fPIC -shared -fopenmp -ffp-contract=off -fno-fast-math -o $@ $< -lm

# Host-side AddressSanitizer / UBSan build (tools/asan_host.py): only the two translation units with host logic are
# instrumented, for the host compilation only; the kernel objects are the ordinary ones.  ~12 minutes (api.hip's device pass).
ASANDIR := build/asan
ASANFLAGS := --offload-arch=$(ARCH) -O1 -g -std=c++17 -ffp-contract=off -fPIC -Iinclude -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer
asan: $(LIB)
	@mkdir -p $(ASANDIR)
	$(HIPCC) $(ASANFLAGS) -c $(CSRC)/api.hip -o $(ASANDIR)/api.o
	$(HIPCC) $(ASANFLAGS) -x hip -c $(CSRC)/comm.cpp -o $(ASANDIR)/comm.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fsanitize=address,undefined -o $(ASANDIR)/libxlbhip_asan.so $(ASANDIR)/api.o $(ASANDIR)/comm.o $(filter-out $(OBJDIR)/api.o $(OBJDIR)/comm.o,$(OBJS)) -ldl

clean:
	rm -rf build xlb_amd/lib/libxlbhip.so oracle/liblbmref.so

.PHONY: all oracle clean asan

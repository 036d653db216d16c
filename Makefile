# Builds the HIP extension in-tree (the .so travels to the GPU box with the repo snapshot).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := montgomery_amd/csrc
LIB := montgomery_amd/libmsm_hip.so
HIPFLAGS := -O3 -pthread -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-function -Wno-unused-variable \
            -Iinclude -I$(CSRC)
BUILD := build
CURVES := CvBls377 CvBls381 CvPallas
CURVE_OBJS := $(CURVES:%=$(BUILD)/kernels_%.o)
HOST_TUS := msm_plan msm_sort msm_tree msm_reduce msm_upload msm_pipeline msm_tables msm_abi msm_test_abi msm_gen sort_kernels te_kernels
HOST_OBJS := $(HOST_TUS:%=$(BUILD)/%.o)
KHDRS := $(CSRC)/msm_kernels.h $(CSRC)/batch_add.h $(CSRC)/msm_gen_kernels.h $(CSRC)/kernel_inst.h $(CSRC)/field.h $(CSRC)/packed.h $(CSRC)/curve.h \
         $(CSRC)/glv.h $(CSRC)/constants_gen.h

# the curve-templated kernels compile once per curve, in parallel with the host pipeline
all:
	$(MAKE) -j8 $(LIB)

$(CSRC)/constants_gen.h: $(CSRC)/gen_constants.py
	python3 $(CSRC)/gen_constants.py

$(BUILD)/kernels_%.o: $(CSRC)/kernels_curve.hip $(KHDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -DMSM_CURVE_TU=$* -c $(CSRC)/kernels_curve.hip -o $@

# the host pipeline is one translation unit per concern (msm_internal.h lists them); the curve-independent kernels have two of their own
HHDRS := $(KHDRS) $(CSRC)/msm_internal.h $(CSRC)/sort_kernels.h $(CSRC)/tree_kernels.h $(CSRC)/te_kernels.h $(CSRC)/host_field.h include/msm_hip.h
$(BUILD)/%.o: $(CSRC)/%.hip $(HHDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(HOST_OBJS) $(CURVE_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -pthread $(HOST_OBJS) $(CURVE_OBJS) -o $(LIB)

clean:
	rm -rf $(LIB) $(BUILD)

.PHONY: all clean

# CPU build of the field / GLV templates for tests/test_host_field.py (no GPU needed)
HOSTTEST := tests/csrc/libfield_host.so
hosttest: $(HOSTTEST)
$(HOSTTEST): tests/csrc/field_host.hip $(CSRC)/field.h $(CSRC)/glv.h $(CSRC)/constants_gen.h
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -I$(CSRC) tests/csrc/field_host.hip -o $(HOSTTEST)

# Node N-API shim (the image's node 12 exposes N-API 8 through /usr/include/node)
NAPI := montgomery_amd/msm_hip.node
napi: $(NAPI)
$(NAPI): napi/msm_addon.cc include/msm_hip.h $(LIB)
	g++ -O2 -std=c++14 -fPIC -shared -Iinclude -I/usr/include/node napi/msm_addon.cc -o $(NAPI) \
	    -Lmontgomery_amd -lmsm_hip -Wl,-rpath,'$$ORIGIN'

# plain-C host of the C ABI (no Python): examples/msm_demo [log2_n] [curve]
demo: examples/msm_demo
examples/msm_demo: examples/msm_demo.c include/msm_hip.h $(LIB)
	gcc -O2 -Wall -Iinclude examples/msm_demo.c -Lmontgomery_amd -lmsm_hip -Wl,-rpath,'$$ORIGIN/../montgomery_amd' -o examples/msm_demo

# A/B experiment builds of the extension: make ab NAME=w2 EXTRA="-DMSM_BA_WAVES=2"  ->  ab_builds/libmsm_w2.so
# (timed against each other by tools/ab_time.py through MSM_HIP_LIB; ab_builds/ is not tracked)
ab:
	@mkdir -p ab_builds/$(NAME)
	for c in $(CURVES); do $(HIPCC) $(HIPFLAGS) $(EXTRA) -DMSM_CURVE_TU=$$c -c $(CSRC)/kernels_curve.hip -o ab_builds/$(NAME)/kernels_$$c.o & done; \
	for t in $(HOST_TUS); do $(HIPCC) $(HIPFLAGS) $(EXTRA) -c $(CSRC)/$$t.hip -o ab_builds/$(NAME)/$$t.o & done; wait
	$(HIPCC) --offload-arch=$(ARCH) -shared -pthread ab_builds/$(NAME)/*.o -o ab_builds/libmsm_$(NAME).so
	rm -rf ab_builds/$(NAME)

# micro-benchmarks quoted in DESIGN.md (run on the GPU box; outputs are committed under profiles/)
UBENCH := ubench_int2 ubench_inv ubench_mul2 ubench_mad3 ubench_gather ubench_carry
ubench:
	for u in $(UBENCH); do $(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -Iinclude -I$(CSRC) tools/$$u.hip -o tools/$$u 2>&1 | grep -E "error" ; done; true

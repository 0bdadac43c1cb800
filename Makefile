# Python-free build of libcusift_amd.so (the HIP kernels + the C ABI of include/cusift_amd.h) for gfx950.
#
#   make                 cusift_amd/libcusift_amd.so  (one object per .hip under build/obj, then one link)
#   make lab             cusift_amd/libcusift_amd_lab.so     (-DCUSIFT_LAB: the tuning overrides of the A/B tools)
#   make stamps          cusift_amd/libcusift_amd_stamps.so  (-DCUSIFT_STAMPS: phase stamps in describe_all_kernel)
#   make oracle          the CPU parity oracle (test infrastructure; never linked by the product)
#   make cpp-tests       the reference's own test programs re-written against include/cuSIFT.h, plain g++
#   make check           runs them (needs a GPU)
#   make install PREFIX=/usr/local    lib/libcusift_amd.so + include/*.h + lib/cmake/cusift_amd/cusift_amdConfig.cmake
#
# The reference builds a static library with CMake (CMakeLists.txt:45-72); CMakeLists.txt next to this file drives this
# same recipe for callers that want find_package(cusift_amd).  `python -m cusift_amd.build` calls `make` too: there is
# one recipe.
HIPCC   ?= $(firstword $(wildcard /opt/rocm/bin/hipcc) hipcc)
ARCH    ?= gfx950
PREFIX  ?= /usr/local
ROOT    := $(abspath $(dir $(lastword $(MAKEFILE_LIST))))
CSRC    := $(ROOT)/cusift_amd/csrc
OBJROOT := $(ROOT)/build/obj

SOURCES := sift_context sift_stages sift_driver sift_stencils sift_keypoints sift_match sift_frontend \
           sift_homography sift_comm sift_tiled sift_pipe
HEADERS := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.inc) $(wildcard $(ROOT)/include/cusift_amd*.h)

# -ffp-contract=off: the only fused multiply-adds are the explicit fmaf() calls (sift_types.h).
# -fno-slp-vectorize: no automatic v_pk_*_f32 packing (measured slower in the keypoint kernel; the blur packs by hand).
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -fno-gpu-rdc -Wall \
            -Wno-unused-function

LIB        := $(ROOT)/cusift_amd/libcusift_amd.so
LIB_LAB    := $(ROOT)/cusift_amd/libcusift_amd_lab.so
LIB_STAMPS := $(ROOT)/cusift_amd/libcusift_amd_stamps.so

.PHONY: all lab stamps ab oracle cpp-tests check install clean
all: $(LIB)
lab: $(LIB_LAB)
stamps: $(LIB_STAMPS)

# $(call variant,<name>,<extra flags>,<library>)
define variant
$(OBJROOT)/$(1)/%.o: $(CSRC)/%.hip $(HEADERS)
	@mkdir -p $$(dir $$@)
	$(HIPCC) $(HIPFLAGS) $(2) -c -o $$@ $$<
$(3): $(addprefix $(OBJROOT)/$(1)/,$(addsuffix .o,$(SOURCES)))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fno-gpu-rdc -o $$@.tmp $$^
	mv -f $$@.tmp $$@
endef
$(eval $(call variant,product,,$(LIB)))
$(eval $(call variant,lab,-DCUSIFT_LAB,$(LIB_LAB)))
$(eval $(call variant,stamps,-DCUSIFT_STAMPS,$(LIB_STAMPS)))

# an experiment's build for a same-box A/B (tools/ab_libs.sh): make ab NAME=x DEFS="-DCUSIFT_SOMETHING" -> tools/ab/x.so
NAME ?= variant
$(eval $(call variant,ab_$(NAME),$(DEFS),$(ROOT)/tools/ab/$(NAME).so))
ab:
	@mkdir -p $(ROOT)/tools/ab
	$(MAKE) -f $(lastword $(MAKEFILE_LIST)) $(ROOT)/tools/ab/$(NAME).so NAME=$(NAME) DEFS="$(DEFS)"

oracle:
	$(MAKE) -C $(ROOT)/oracle

cpp-tests: $(LIB)
	$(MAKE) -C $(ROOT)/tests/cpp

check: cpp-tests
	$(MAKE) -C $(ROOT)/tests/cpp check

install: $(LIB)
	install -d $(DESTDIR)$(PREFIX)/lib $(DESTDIR)$(PREFIX)/include/cusift_amd $(DESTDIR)$(PREFIX)/lib/cmake/cusift_amd
	install -m 0755 $(LIB) $(DESTDIR)$(PREFIX)/lib/
	install -m 0644 $(ROOT)/include/*.h $(DESTDIR)$(PREFIX)/include/cusift_amd/
	install -m 0644 $(ROOT)/cmake/cusift_amdConfig.cmake $(DESTDIR)$(PREFIX)/lib/cmake/cusift_amd/

clean:
	rm -rf $(ROOT)/build/obj $(LIB) $(LIB_LAB) $(LIB_STAMPS)

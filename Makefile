# Builds the gfx950 shared library (C ABI in include/arrowspace_hip.h) and the CPU checker.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := pyarrowspace_amd/csrc
SRCS := $(CSRC)/as_api.hip $(CSRC)/as_build.hip $(CSRC)/as_k2bf.hip $(CSRC)/as_comm.hip $(CSRC)/as_edges.hip $(CSRC)/as_feat.hip $(CSRC)/as_scan.hip $(CSRC)/as_search.hip
HDRS := $(CSRC)/as_common.hpp $(CSRC)/as_query.hpp $(CSRC)/as_knn.hpp include/arrowspace_hip.h
OBJS := $(SRCS:.hip=.o)
LIB := pyarrowspace_amd/libarrowspace_hip.so
ABLATION ?= 0
STAMPS ?= 0
HIPFLAGS ?= $(if $(filter 1,$(ABLATION)),-DAS_ABLATION) $(if $(filter 1,$(STAMPS)),-DAS_STAMPS) -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -Wall -Wno-unused-function -Wno-unused-value

all: $(LIB) oracle
$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl
oracle:
	$(MAKE) -C oracle -s
clean:
	rm -f $(OBJS) $(LIB)
	$(MAKE) -C oracle clean
.PHONY: all oracle clean

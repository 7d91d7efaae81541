# Top-level build.  `make` builds the product library (HIP, gfx950) and the
# oracle (test infrastructure); `make ref` builds oracle/_ref from /root/reference
# when that tree is present (this container only).
HIPCC ?= /opt/rocm/bin/hipcc
CXX   ?= g++
ARCH  ?= gfx950

CSRC := plade_amd/csrc
HIP_SRCS := $(wildcard $(CSRC)/*.hip)
HIP_OBJS := $(patsubst $(CSRC)/%.hip,build/%.o,$(HIP_SRCS))
HIP_HDRS := $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/*.cuh) include/plade_hip.h
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(CSRC) -Wno-unused-result

all: lib oracle cli

lib: plade_amd/libplade_hip.so
oracle: oracle/libplade_oracle.so
cli: plade_amd/PLADE

build/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

# host-only parts of the library (the PLY ingest: plade_ply_read of include/plade_hip.h and the CLI share it)
build/ply_reader.o: $(CSRC)/ply_reader.cpp $(CSRC)/ply_reader.h include/plade_hip.h
	@mkdir -p build
	$(CXX) -O2 -std=c++17 -fPIC -Iinclude -I$(CSRC) -c $< -o $@

plade_amd/libplade_hip.so: $(HIP_OBJS) build/ply_reader.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

HOST_SRCS := $(CSRC)/main.cpp $(CSRC)/plade_host.cpp
plade_amd/PLADE: $(HOST_SRCS) $(CSRC)/plade.h $(CSRC)/plade_compat.h $(CSRC)/ply_reader.h plade_amd/libplade_hip.so
	$(CXX) -O2 -std=c++17 -pthread -Iinclude -I$(CSRC) $(HOST_SRCS) -o $@ -Lplade_amd -lplade_hip -Wl,-rpath,'$$ORIGIN'

oracle/libplade_oracle.so: oracle/plade_oracle.cpp oracle/plade_oracle.h oracle/orc_math.h
	$(CXX) -O2 -std=c++14 -fPIC -ffp-contract=off -shared -o $@ oracle/plade_oracle.cpp

ref:
	@if [ -d /root/reference/code/3rd_party ]; then $(MAKE) -C oracle/ref -j8; else echo "no /root/reference: using prebuilt oracle/_ref"; fi

clean:
	rm -rf build plade_amd/libplade_hip.so plade_amd/PLADE oracle/libplade_oracle.so
.PHONY: all lib oracle cli ref clean

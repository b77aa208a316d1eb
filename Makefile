# Build of the MI355X-native Archon BWT path.
#   make lib      -> dark-archon_amd/libarchon_hip.so   (HIP kernels + C ABI, gfx950)
#   make cli      -> bin/archon                          (C++ host: Archon class mirror + a7 CLI)
#   make oracle   -> oracle/liboracle.so (+ oracle/_ref/* when /root/reference is present)
HIPCC     ?= hipcc
CXX       ?= g++
ARCH      ?= gfx950
PKG        = dark-archon_amd
CSRC       = $(PKG)/csrc
HIPFLAGS   = -O3 --offload-arch=$(ARCH) -std=c++17 -fPIC -Wall -Wno-unused-function
LIB        = $(PKG)/libarchon_hip.so

all: lib host cli oracle

lib: $(LIB)

$(LIB): $(CSRC)/archon_hip.hip $(wildcard $(CSRC)/*.hiph) include/archon_hip.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/archon_hip.hip

host: $(PKG)/libarchon.so

# the block-coder object (include/archon.h) as a shared library for FFI callers
$(PKG)/libarchon.so: $(PKG)/host/archon_host.cpp $(PKG)/host/archon_post.cpp $(PKG)/host/archon_host.h include/archon.h $(LIB)
	$(CXX) -O2 -std=c++17 -Wall -fPIC -shared -Iinclude -o $@ $(PKG)/host/archon_host.cpp $(PKG)/host/archon_post.cpp \
	    -L$(PKG) -larchon_hip -Wl,-rpath,'$$ORIGIN'

cli: bin/archon

bin/archon: $(PKG)/host/archon_host.cpp $(PKG)/host/archon_main.cpp $(PKG)/host/archon_container.cpp $(PKG)/host/archon_post.cpp $(PKG)/host/archon_host.h include/archon.h $(LIB)
	@mkdir -p bin
	$(CXX) -O2 -std=c++17 -Wall -pthread -Iinclude -o $@ $(PKG)/host/archon_main.cpp $(PKG)/host/archon_host.cpp $(PKG)/host/archon_container.cpp $(PKG)/host/archon_post.cpp \
	    -L$(PKG) -larchon_hip -Wl,-rpath,'$$ORIGIN/../$(PKG)'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf bin $(LIB) $(PKG)/libarchon.so
	$(MAKE) -C oracle clean

.PHONY: all lib host cli oracle clean micro

# micro-benchmarks and test-only kernels behind the measurements quoted in DESIGN.md (not part of the product)
micro: tools/micro/scatter_bw tools/micro/scatter_align tools/micro/libcu_hog.so
tools/micro/scatter_bw: tools/micro/scatter_bw.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) -o $@ $<
tools/micro/scatter_align: tools/micro/scatter_align.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) -o $@ $<
tools/micro/libcu_hog.so: tools/micro/cu_hog.hip
	$(HIPCC) -O3 --offload-arch=$(ARCH) -fPIC -shared -o $@ $<

# experiments build for tools/ (phase stamps etc.); never loaded by the product, the tests or bench.py
exp: $(PKG)/libarchon_hip_exp.so
$(PKG)/libarchon_hip_exp.so: $(CSRC)/archon_hip.hip $(wildcard $(CSRC)/*.hiph) include/archon_hip.h
	$(HIPCC) $(HIPFLAGS) -DARCHON_EXPERIMENTS $(EXPFLAGS) -shared -o $@ $(CSRC)/archon_hip.hip

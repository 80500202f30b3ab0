# Build without Python: libkmu.so (hipcc, gfx950), the C++ tools and the C example (g++ / gcc against the C-ABI).
# `python -c "import __graft_entry__ as g; g.build()"` does the same and also builds the test infrastructure.
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
CC       ?= gcc
HIPFLAGS ?= --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function
CSRC     := kmerutils_amd/csrc
OBJDIR   := kmerutils_amd/build
SOURCES  := kmu_api kmu_sketch kmu_sketch_kernels kmu_sketch_super kmu_sketch_dens kmu_count kmu_count_part kmu_count_dist kmu_smer kmu_hostpack kmu_compare kmu_ingest kmu_kmergen kmu_comm
OBJS     := $(SOURCES:%=$(OBJDIR)/%.o)
LIB      := kmerutils_amd/libkmu.so
BIN      := kmerutils_amd/bin
LINK     := -Lkmerutils_amd -lkmu -Wl,-rpath-link,/opt/rocm/lib

all: $(LIB) $(BIN)/datasketcher $(BIN)/parsefastq examples/sketch_c

$(OBJDIR)/%.o: $(CSRC)/%.hip $(CSRC)/kmu_device.h $(CSRC)/kmu_stream.h $(CSRC)/kmu_ctx.hpp $(CSRC)/kmu_comm.hpp $(CSRC)/kmu_flat.h $(CSRC)/kmu_count_table.h $(CSRC)/kmu_sketch_kernels.h $(CSRC)/kmu_smer.h $(CSRC)/kmu_smer.hpp $(CSRC)/kmu_hostpack.hpp include/kmu.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(OBJS) -ldl -lpthread

$(BIN)/%: kmerutils_amd/tools/%.cpp include/kmerutils.hpp include/kmu.h $(LIB)
	@mkdir -p $(BIN)
	$(CXX) -std=c++17 -O2 -Wall -Wextra -pthread $< -o $@ $(LINK) -Wl,-rpath,'$$ORIGIN/..'

examples/sketch_c: examples/sketch.c include/kmu.h $(LIB)
	$(CC) -std=c99 -O2 -Wall -Wextra $< -o $@ $(LINK) -Wl,-rpath,'$$ORIGIN/../kmerutils_amd'

# CPU-side sanitizer pass (no GPU): the oracle's C restatement and the host-only code of include/kmerutils.hpp under
# AddressSanitizer + UndefinedBehaviorSanitizer -- the host-only C++ test program, then the whole `-m "not gpu"` suite with the
# sanitized oracle preloaded into python.  GPU AddressSanitizer is not available on this pool.
SANFLAGS := -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer
sanitize:
	@mkdir -p oracle/_build/san tests/cpp/_build/san
	$(CC) -std=c99 -fPIC $(SANFLAGS) -ffp-contract=off -fno-fast-math -Wall -Wextra -shared -o oracle/_build/san/libkmu_oracle.so oracle/kmu_oracle.c -lm
	$(CXX) -std=c++17 $(SANFLAGS) -Wall -Wextra -pthread tests/cpp/test_host_san.cpp -o tests/cpp/_build/san/test_host_san
	ASAN_OPTIONS=detect_leaks=1 tests/cpp/_build/san/test_host_san tests/cpp/_build/san
	KMU_ORACLE_SO=$(CURDIR)/oracle/_build/san/libkmu_oracle.so LD_PRELOAD=$$($(CC) -print-file-name=libasan.so):$$($(CC) -print-file-name=libubsan.so) \
	  ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider

clean:
	rm -rf $(OBJDIR) $(LIB) $(BIN) examples/sketch_c oracle/_build/san tests/cpp/_build/san

.PHONY: all clean sanitize

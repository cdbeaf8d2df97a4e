# Build recipe for the gfx950 radix-sort engine, the CPU oracle and the harness.
#   make            -> radx_amd/lib/libradx_hip.so, oracle/libradx_oracle.so, harness/sort_test
#   make resources  -> per-kernel VGPR/SGPR/LDS/occupancy report
HIPCC    ?= /opt/rocm/bin/hipcc
CC       ?= gcc
CXX      ?= g++
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Iinclude -Wall -Wno-unused-function

LIB      := radx_amd/lib/libradx_hip.so
ORACLE   := oracle/libradx_oracle.so
CPUSORT  := oracle/libradx_cpusort.so
HARNESS  := harness/sort_test
SHARDED  := harness/sharded_test
REFSEQ   := harness/reference_sequence

all: $(LIB) $(ORACLE) $(CPUSORT) $(HARNESS) $(SHARDED) $(REFSEQ)

$(LIB): radx_amd/csrc/radx_hip.hip include/radx_hip.h
	@mkdir -p radx_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ radx_amd/csrc/radx_hip.hip

$(ORACLE): oracle/radx_oracle.c oracle/radx_oracle.h
	$(CC) -O2 -std=c11 -fPIC -shared -Wall -Wextra -o $@ oracle/radx_oracle.c

$(CPUSORT): oracle/cpu_sort.cpp
	$(CXX) -O3 -std=c++17 -fPIC -shared -fopenmp -Wall -o $@ oracle/cpu_sort.cpp

$(HARNESS): harness/sort_test.cpp $(LIB) $(wildcard include/radx/*.hpp) include/radx_hip.h
	$(CXX) -O2 -std=c++17 -fopenmp -Wall -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
	    -o $@ harness/sort_test.cpp \
	    -Lradx_amd/lib -lradx_hip -L/opt/rocm/lib -lamdhip64 \
	    -Wl,-rpath,'$$ORIGIN/../radx_amd/lib' -Wl,-rpath,/opt/rocm/lib

# the reference harness's call sequence against these headers (compile check; on a GPU it also runs)
$(REFSEQ): harness/reference_sequence.cpp $(LIB) $(wildcard include/radx/*.hpp) include/radx_hip.h
	$(CXX) -O2 -std=c++17 -Wall -Wextra -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
	    -o $@ harness/reference_sequence.cpp \
	    -Lradx_amd/lib -lradx_hip -L/opt/rocm/lib -lamdhip64 \
	    -Wl,-rpath,'$$ORIGIN/../radx_amd/lib' -Wl,-rpath,/opt/rocm/lib

# the multi-GPU harness links RCCL; it prefers the RCCL that ships with PyTorch when present at
# run time only through LD_LIBRARY_PATH — the build uses ROCm's own
$(SHARDED): harness/sharded_test.cpp $(LIB) $(wildcard include/radx/*.hpp) include/radx_hip.h
	$(CXX) -O2 -std=c++17 -pthread -Wall -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
	    -o $@ harness/sharded_test.cpp \
	    -Lradx_amd/lib -lradx_hip -L/opt/rocm/lib -lamdhip64 -lrccl \
	    -Wl,-rpath,'$$ORIGIN/../radx_amd/lib' -Wl,-rpath,/opt/rocm/lib

# Host-side sanitizer builds (test infrastructure): the oracle + CPU comparator behind a driver
# that runs every geometry and edge size on exact-size heap buffers, and the harness (its
# --cpu-only leg touches no GPU).  tests/test_sanitizers.py runs both and fails on any report.
SANFLAGS := -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1
asan: oracle/asan_driver harness/sort_test_asan
oracle/asan_driver: oracle/asan_driver.cpp oracle/radx_oracle.c oracle/radx_oracle.h oracle/cpu_sort.cpp
	$(CC) $(SANFLAGS) -std=c11 -Wall -Wextra -c -o oracle/radx_oracle_asan.o oracle/radx_oracle.c
	$(CXX) $(SANFLAGS) -std=c++17 -fopenmp -Wall -Ioracle -o $@ oracle/asan_driver.cpp oracle/cpu_sort.cpp oracle/radx_oracle_asan.o
harness/sort_test_asan: harness/sort_test.cpp $(LIB) $(wildcard include/radx/*.hpp) include/radx_hip.h
	$(CXX) $(SANFLAGS) -std=c++17 -fopenmp -Wall -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
	    -o $@ harness/sort_test.cpp \
	    -Lradx_amd/lib -lradx_hip -L/opt/rocm/lib -lamdhip64 \
	    -Wl,-rpath,'$$ORIGIN/../radx_amd/lib' -Wl,-rpath,/opt/rocm/lib

resources:
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c -o /dev/null radx_amd/csrc/radx_hip.hip 2>&1 | \
	    grep -E "Function Name|VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize" | grep -v "^$$" || true

# development micro-benchmark: HBM copy ceilings by access width and launch shape
micro: tools/micro/copy_bench tools/micro/rocprim_sort_bench tools/micro/dispatch_bench
tools/micro/dispatch_bench: tools/micro/dispatch_bench.hip
	$(HIPCC) -O3 -w --offload-arch=$(ARCH) -o $@ $<
tools/micro/rocprim_sort_bench: tools/micro/rocprim_sort_bench.hip
	$(HIPCC) -O3 -w --offload-arch=$(ARCH) -o $@ $<
tools/micro/copy_bench: tools/micro/copy_bench.hip
	$(HIPCC) -O3 -w --offload-arch=$(ARCH) -o $@ $<

clean:
	rm -f $(LIB) $(ORACLE) $(CPUSORT) $(HARNESS) $(SHARDED) $(REFSEQ) tools/micro/copy_bench tools/micro/rocprim_sort_bench \
	    oracle/asan_driver oracle/radx_oracle_asan.o harness/sort_test_asan

.PHONY: all clean resources micro asan

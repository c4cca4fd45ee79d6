# Builds everything that __graft_entry__.build() builds: the HIP libraries for gfx950, the CPU oracle (test
# infrastructure), the probes compiled against the reference tree where it is mounted, and the C++ test programs.
REF ?= /root/reference

all:
	$(MAKE) -C ray-tracing_amd/csrc
	$(MAKE) -C oracle oracle
	@if grep -q " fma" /proc/cpuinfo; then $(MAKE) -C oracle variant; fi
	@if [ -d $(REF)/src/rt ]; then $(MAKE) -C oracle ref REF=$(REF); fi
	$(MAKE) -C tests

test: all
	python -m pytest tests -q -m "not gpu"

clean:
	$(MAKE) -C ray-tracing_amd/csrc clean
	$(MAKE) -C oracle clean
	$(MAKE) -C tests clean

.PHONY: all test clean

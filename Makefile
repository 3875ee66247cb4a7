# Convenience targets (the driver uses __graft_entry__.build() / pytest / bench.py directly).
PY ?= python

.PHONY: build test test-gpu bench smoke golden demo clean

build:            ## hipcc --offload-arch=gfx950 -> keep_amd/libkeep_hip.so (cross-compiles without a GPU)
	$(PY) -m keep_amd.build

test: build       ## oracle vs golden vectors, ABI / layout / sharding checks (no GPU)
	$(PY) -m pytest tests -q -m "not gpu"

test-gpu: build   ## HIP path vs oracle (needs an MI355X)
	$(PY) -m pytest tests -q -m gpu

bench: build      ## one JSON line: tiles/s + roofline + parity + CPU baseline
	$(PY) bench.py

smoke: build
	$(PY) __graft_entry__.py --smoke

golden:           ## regenerate tests/golden/*.npz (needs /root/reference; build container only)
	$(PY) tools/make_golden.py

demo: build       ## the C ABI from plain C++
	hipcc -O2 -Iinclude examples/c_abi_demo.cpp -Lkeep_amd -lkeep_hip -Wl,-rpath,$(CURDIR)/keep_amd -o examples/c_abi_demo

clean:
	rm -rf keep_amd/build keep_amd/libkeep_hip.so examples/c_abi_demo

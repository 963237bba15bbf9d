#!/bin/bash
# k_t_block's phases (tools/exp_t_stamps.py). Run on the GPU box from the repo root; the stamped library is built HERE first:
#   cd pagraph_amd/csrc && mkdir -p build_stamps && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DPG_T_STAMPS -c pg_sample.hip -o build_stamps/pg_sample.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpagraph_hip_stamps.so $(ls build/*.o | grep -v pg_sample.o) build_stamps/pg_sample.o -lpthread -L/opt/rocm/lib -lhsa-runtime64
cp pagraph_amd/libpagraph_hip_stamps.so pagraph_amd/libpagraph_hip.so      # (the box's copy of the repo is thrown away)
timeout 600 python tools/exp_t_stamps.py

#!/bin/bash
# builds tools/pmc_calib/pmc_calib for gfx950 (cross-compiles without a GPU); the binary is git-ignored and travels with the gpurun snapshot
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pmc_calib pmc_calib.hip

#!/bin/bash
# the GPU suite with a readable tail (RCCL prints a five-line banner at exit): bash tools/gpu_suite.sh [pytest args]
# (the whole log goes to gpurun_out/suite_full.txt)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -X faulthandler -m pytest tests -m gpu -q "$@" > gpurun_out/suite_full.txt 2>&1
grep -v -e "^RCCL version" -e "^HIP version" -e "^ROCm version" -e "^Hostname" -e "^Librccl path" -e "amdgpu.ids" gpurun_out/suite_full.txt | tail -25

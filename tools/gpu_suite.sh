#!/bin/bash
# the GPU suite with a readable tail (RCCL prints a five-line banner at exit): bash tools/gpu_suite.sh [pytest args]
cd "$(dirname "$0")/.."
python -m pytest tests -m gpu -q "$@" 2>&1 | grep -v -e "^RCCL version" -e "^HIP version" -e "^ROCm version" -e "^Hostname" -e "^Librccl path" -e "amdgpu.ids" | tail -25

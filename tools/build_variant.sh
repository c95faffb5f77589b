#!/bin/bash
# Development aid: build a variant of the library, e.g. tools/build_variant.sh stamps -DMFM_STAMPS -DMFM_EXP_PRIO
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
cd $R/mfm_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -mllvm -amdgpu-sched-strategy=${MFM_SCHED:-max-ilp} -mllvm -amdgpu-kernarg-preload-count=${MFM_PRELOAD:-12} "$@" -o $R/mfm_amd/lib/libmfm_hip_$name.so api.hip
echo built $R/mfm_amd/lib/libmfm_hip_$name.so

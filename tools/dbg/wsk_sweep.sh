#!/bin/bash
# Development aid: tools/iter_time.py over decompositions of wgrad_sk (MFM_WSK_G, MFM_WSK_XCD) and library variants, on one box.
cd "$(dirname "$0")/../.."
run() { echo "== $*"; env "$@" timeout -k 10 120 python tools/iter_time.py 2>&1 | grep -v amdgpu.ids | tail -n 3 | head -n 2; }
run MFM_WSK_G=416
run MFM_LIB=$PWD/mfm_amd/lib/libmfm_hip_A.so
run MFM_WSK_G=256
run MFM_WSK_G=256 MFM_WSK_XCD=1
run MFM_WSK_G=512 MFM_WSK_XCD=1
run MFM_WSK_G=208
run MFM_WSK_G=256 MFM_LIB=$PWD/mfm_amd/lib/libmfm_hip_ns6.so
run MFM_WSK_G=256 MFM_WSK_XCD=1 MFM_LIB=$PWD/mfm_amd/lib/libmfm_hip_ns6.so

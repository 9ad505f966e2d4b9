#!/bin/bash
# Developer A/B on one box: dev builds of conv_split_kernels.hip (-DSSTEM_SPLIT_DEV=1 [-DSSTEM_SPLIT_ABLATE=mask]) linked into build_ablate/libsstem_dev_<name>.so
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for m in ${ABL:-0 commit tab}; do echo -n "$m: "; SSTEM_NATIVE_LIB=$PWD/build_ablate/libsstem_dev_$m.so python tools/time_conv.py f16x3 8,32,1024,1024,32 8,64,1024,1024,32 8,32,512,512,32 8,64,512,512,64 8,256,128,128,256 8,51,1024,1024,51 2>&1 | grep -v amdgpu.ids; done; done

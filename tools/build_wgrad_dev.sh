#!/bin/bash
# Developer builds of the split weight-gradient kernel alone (seconds): tools/build_wgrad_dev.sh <name> "<hipcc -D flags>"
#   -> build_ablate/libsstem_wgrad_<name>.so (the other objects come from the product build: run make first); select with SSTEM_NATIVE_LIB=
#   e.g. tools/build_wgrad_dev.sh stamps "-DSSTEM_WGRAD_STAMPS=1"   (tools/wgrad_stamps.py prints the phase times)
set -e
cd "$(dirname "$0")/../sstem-restoration_amd/csrc"
OUT=../../build_ablate
mkdir -p $OUT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $2 -c conv_split_wgrad.hip -o $OUT/conv_split_wgrad_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsstem_wgrad_$1.so $OUT/conv_split_wgrad_$1.o sstem_capi.o sepconv_kernels.o conv_kernels.o \
    conv_bf16_kernels.o conv_split_kernels.o convt_kernels.o warp_kernels.o misc_kernels.o norm_kernels.o

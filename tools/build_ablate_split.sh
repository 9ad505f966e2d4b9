#!/bin/bash
# Developer tool: ablation builds of the split-bf16 convolution kernels into build_ablate/libsstem_split_<mask>.so
#   tools/build_ablate_split.sh <mask>      (SSTEM_SPLIT_ABLATE bits: top of conv_split_kernels.hip; results are wrong by design)
#   tools/build_ablate_split.sh notail      (A/B build without the tap-row last chunk: -DSSTEM_SPLIT_TAIL=0; results are right)
# Select one at run time with SSTEM_NATIVE_LIB=<path>.  The other objects come from the product build (run make first).
set -e
cd "$(dirname "$0")/../sstem-restoration_amd/csrc"
OUT=../../build_ablate
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
DEF="-DSSTEM_SPLIT_ABLATE=$1"
if [ "$1" = "notail" ]; then DEF="-DSSTEM_SPLIT_TAIL=0"; fi
if [ -n "$2" ]; then DEF="$2"; fi          # tools/build_ablate_split.sh <name> "<flags>"
/opt/rocm/bin/hipcc $FLAGS $DEF -c conv_split_kernels.hip -o $OUT/conv_split_$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsstem_split_$1.so $OUT/conv_split_$1.o sstem_capi.o sepconv_kernels.o conv_kernels.o \
    conv_bf16_kernels.o convt_kernels.o warp_kernels.o misc_kernels.o norm_kernels.o

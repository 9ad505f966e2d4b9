cd $GRAFT_REPO_ROOT
for rep in 1 2; do for m in 0 1 2 4 8 64 15; do echo -n "ablate=$m: "; SSTEM_NATIVE_LIB=$PWD/build_ablate/libsstem_split_dev$m.so python tools/time_conv.py f16x3 8,64,512,512,64 8,128,256,256,128 8,256,128,128,256 8,64,1024,1024,32 2>&1 | grep -v amdgpu.ids; done; done

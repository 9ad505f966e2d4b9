for rep in 1 2; do
echo -n "product-lib: "; python tools/time_conv.py f16x3 8,6,1024,1024,32 8,6,1024,1024,6 8,32,1024,1024,32 8,48,512,512,32 2>&1 | grep -v amdgpu
echo -n "odd-cpk dev: "; SSTEM_NATIVE_LIB=$PWD/build_ablate/libsstem_dev_odd.so python tools/time_conv.py f16x3 8,6,1024,1024,32 8,6,1024,1024,6 8,32,1024,1024,32 8,48,512,512,32 2>&1 | grep -v amdgpu
done

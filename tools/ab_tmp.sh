mkdir -p gpurun_out/r3s
L="8,64,512,512,64 8,32,1024,1024,32 8,256,128,128,256 8,128,256,256,128 8,51,1024,1024,51"
O=gpurun_out/r3s/f16_kxo.txt
: > $O
timeout -k 10 900 python -m pytest tests/test_conv_f16x3_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $O
for i in 1 2; do
timeout -k 10 120 python tools/time_conv.py f16x3 $L >> $O 2>/dev/null &&
SSTEM_F16_KXO=0 timeout -k 10 120 python tools/time_conv.py f16x3 $L >> $O 2>/dev/null || exit 1
done
timeout -k 10 200 python tools/bench_models.py --what ifnet --iters 20 2>/dev/null | grep "IFNet forward" >> $O
SSTEM_F16_KXO=0 timeout -k 10 200 python tools/bench_models.py --what ifnet --iters 20 2>/dev/null | grep "IFNet forward" >> $O
cat $O

mkdir -p gpurun_out/r3s
O=gpurun_out/r3s/prefetch_flow.txt
: > $O
for b in 2 16; do
timeout -k 10 200 python tools/bench_models.py --what fusion_step --fusion-batch $b --graph --iters 100 2>/dev/null | grep "fusion step" >> $O &&
timeout -k 10 200 python tools/bench_models.py --what fusion_step --fusion-batch $b --graph --prefetch-flow --iters 100 2>/dev/null | grep "fusion step" >> $O || exit 1
done
timeout -k 10 200 python tools/bench_models.py --what fusion_step --fusion-batch 2 --iters 100 2>/dev/null | grep "fusion step" >> $O
timeout -k 10 200 python tools/bench_models.py --what fusion_step --fusion-batch 2 --prefetch-flow --iters 100 2>/dev/null | grep "fusion step" >> $O
cat $O
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "c3" 2>&1 | tail -5

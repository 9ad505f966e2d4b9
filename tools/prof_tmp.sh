mkdir -p gpurun_out/r3s/prof
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3s/prof/bench -- python3 $R/bench.py --no-cpu-baseline --no-extra > $R/gpurun_out/r3s/prof/bench_line.json 2> $R/gpurun_out/r3s/prof/bench.err || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3s/prof/ifnet -- python3 $R/tools/bench_models.py --what ifnet --iters 10 > $R/gpurun_out/r3s/prof/ifnet.txt 2> $R/gpurun_out/r3s/prof/ifnet.err || exit 1
cd $R
python3 tools/kernel_stats_top.py gpurun_out/r3s/prof/bench > gpurun_out/r3s/prof/bench_top.txt 2>&1
python3 tools/kernel_stats_top.py gpurun_out/r3s/prof/ifnet > gpurun_out/r3s/prof/ifnet_top.txt 2>&1
find gpurun_out/r3s/prof -name "*kernel_stats.csv" | head
find gpurun_out/r3s/prof -name "*_kernel_trace.csv" -delete
find gpurun_out/r3s/prof -name "*.db" -delete
head -12 gpurun_out/r3s/prof/ifnet_top.txt; head -5 gpurun_out/r3s/prof/bench_top.txt; tail -c 400 gpurun_out/r3s/prof/bench_line.json

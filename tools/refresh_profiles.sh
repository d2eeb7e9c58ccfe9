set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_B.json 2> gpurun_out/final/bench_B.err
python bench.py --config C --steps 100 --warmup 10 --cpu-seconds 6 > gpurun_out/final/bench_C.json 2> gpurun_out/final/bench_C.err
python bench.py --config E --steps 30 --warmup 3 --cpu-seconds 6 > gpurun_out/final/bench_E.json 2> gpurun_out/final/bench_E.err
python bench.py --host-api --cpu-seconds 0 > gpurun_out/final/bench_B_hostapi.json 2> gpurun_out/final/bench_B_hostapi.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/final/prof --output-format csv -- python3 bench.py --steps 200 --warmup 20 --cpu-seconds 0 > gpurun_out/final/prof_bench.json 2> gpurun_out/final/prof.err
find gpurun_out/final/prof -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/final/kernel_stats.csv
bash tools/pmc_profile.sh gpurun_out/final/pmc > gpurun_out/final/pmc.log 2>&1
tail -3 gpurun_out/final/bench_B.json

set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 900 python bench.py ) > gpurun_out/bench_default.log 2>&1
tail -5 gpurun_out/bench_default.log
timeout 600 python bench.py --profile-all --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/bench_profall.log 2>&1
tail -3 gpurun_out/bench_profall.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1d -o r1d -- python bench.py --no-cpu-baseline > gpurun_out/bench_rocprof.log 2>&1
tail -2 gpurun_out/bench_rocprof.log
rm -f gpurun_out/prof_r1d/*kernel_trace.csv gpurun_out/prof_r1d/*.db
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --no-cpu-baseline --no-graph --no-roofline --steps 2 --warmup 1 > gpurun_out/pmc_fetch.log 2>&1
tail -2 gpurun_out/pmc_fetch.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --no-cpu-baseline --no-graph --no-roofline --steps 2 --warmup 1 > gpurun_out/pmc_write.log 2>&1
tail -2 gpurun_out/pmc_write.log
python tools/pmc_summary.py gpurun_out/pmc_fetch/f_counter_collection.csv FETCH_SIZE > gpurun_out/pmc_fetch_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/pmc_write/w_counter_collection.csv WRITE_SIZE > gpurun_out/pmc_write_summary.txt 2>&1
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
du -sh gpurun_out

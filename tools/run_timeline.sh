# usage: bash tools/run_timeline.sh <out.txt> [bench flags...]   -- rocprof kernel trace of a graph-replayed step -> per-kernel table
OUT=${1:-gpurun_out/timeline.txt}; shift
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python bench.py --no-cpu-baseline --no-roofline --steps 5 "$@" > gpurun_out/tl.log 2>&1
tail -1 gpurun_out/tl.log | cut -c1-200
python tools/timeline.py gpurun_out/tl/tl_kernel_trace.csv $OUT ${OUT%.txt}_sequence.txt
rm -rf gpurun_out/tl

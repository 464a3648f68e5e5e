# usage: ab_lib.sh OTHER.so [bench flags...] -- bench with the in-tree library and with OTHER.so, interleaved, on the same box
other=$1; shift
for i in 1 2; do
for lib in "" "$other"; do
echo "lib=${lib:-default}"; TOIST_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-roofline "$@" 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done

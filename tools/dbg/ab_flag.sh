# usage: ab_flag.sh "<flags A>" "<flags B>" -- bench with either flag set, interleaved twice, on the same box
for i in 1 2; do
for f in "$1" "$2"; do
echo "flags: $f"; python bench.py --no-cpu-baseline --no-roofline $f 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done

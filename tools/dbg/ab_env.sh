# usage: ab_env.sh VAR A B  -- bench twice with VAR=A and VAR=B, interleaved, on the same box
for i in 1 2; do
for t in "$2" "$3"; do
echo "$1=$t"; env $1=$t python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done

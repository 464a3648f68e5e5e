# usage: ab_multi.sh "ENV1=a ENV2=b" "ENV1=c" ...  -- bench under each environment, two interleaved rounds, same box
for i in 1 2; do
for e in "$@"; do
echo "env: $e"; env $e python bench.py --no-cpu-baseline --no-roofline 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done

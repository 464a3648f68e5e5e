# dynamic instruction mix per kernel over two eager training steps (rocprofv3 PMC pass)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_i -o c -- python bench.py --no-cpu-baseline --no-graph --no-roofline --steps 2 --warmup 1 > gpurun_out/pmc_i.log 2>&1
tail -2 gpurun_out/pmc_i.log | cut -c1-200
python - <<'PY'
import csv, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open('gpurun_out/pmc_i/c_counter_collection.csv')):
    n=r['Kernel_Name']
    n=re.sub(r'\(anonymous namespace\)::','',n).replace('void toist::','').replace('toist::','')[:60]
    acc[n][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[n]+=1
rows=sorted(acc.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES'])
print("%-62s %6s %10s %8s %8s %8s %8s %7s" % ("kernel", "calls", "wavecyc(M)", "VALU/w", "SALU/w", "LDS/w", "MFMA/w", "wait%"))
for n,v in rows[:40]:
    w=max(v['SQ_WAVES'],1)
    print("%-62s %6d %10.1f %8.0f %8.0f %8.0f %8.0f %6.0f%%" % (n, cnt[n], v['SQ_WAVE_CYCLES']/1e6, v['SQ_INSTS_VALU']/w, v['SQ_INSTS_SALU']/w, v['SQ_INSTS_LDS']/w, v['SQ_INSTS_MFMA']/w, 100*v['SQ_WAIT_ANY']/max(v['SQ_WAVE_CYCLES'],1)))
PY
rm -rf gpurun_out/pmc_i

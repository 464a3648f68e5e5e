# dynamic instruction mix of the conv / GEMM kernels (rocprofv3 PMC pass over tools/bench_conv3.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_i -o c -- python tools/bench_conv3.py > gpurun_out/pmc_i.log 2>&1
tail -3 gpurun_out/pmc_i.log
python - <<'PY'
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open('gpurun_out/pmc_i/c_counter_collection.csv')):
    n=r['Kernel_Name']
    if 'conv3' in n or 'gemm_kernel' in n:
        key=(n.replace('void toist::','')[:48], r['Grid_Size'])
        acc[key][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[key]+=1
for k,v in sorted(acc.items()):
    n=cnt[k]; w=v['SQ_WAVES']/n
    print(k, 'waves', int(w), {c: round(x/n/w) for c,x in v.items() if c!='SQ_WAVES'})
PY
rm -rf gpurun_out/pmc_i

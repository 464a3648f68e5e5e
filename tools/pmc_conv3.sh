cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_c3 -o c -- python tools/bench_conv3.py > gpurun_out/pmc_c3.log 2>&1
python - <<'PY'
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open('gpurun_out/pmc_c3/c_counter_collection.csv')):
    n=r['Kernel_Name']
    if 'conv3' in n or 'gemm_kernel<64, 64, 64, 2, 0' in n or 'gemm_kernel<64, 64, 64, 3, 1' in n:
        key=(n[:60], r['Grid_Size'])
        acc[key][r['Counter_Name']]+=float(r['Counter_Value']); 
        if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[key]+=1
for k,v in acc.items():
    n=cnt[k]
    print(k, n, {c: round(x/n) for c,x in v.items()})
PY
rm -rf gpurun_out/pmc_c3

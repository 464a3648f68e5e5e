# dynamic instruction mix / stall split of the attention-core kernels (tools/bench_attn_core.py) -- rocprofv3 PMC passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for SET in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAVES"; do
rm -rf gpurun_out/pmc_a
timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/pmc_a -o c -- python tools/bench_attn_core.py > gpurun_out/pmc_a.log 2>&1
python - <<'PY'
import csv, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open('gpurun_out/pmc_a/c_counter_collection.csv')):
    n=r['Kernel_Name']
    if 'attn' not in n and 'softmax' not in n: continue
    n=re.sub(r'\(anonymous namespace\)::','',n).replace('void toist::','').replace('toist::','')[:34]
    key=(n, r.get('Grid_Size','?'), r.get('LDS_Block_Size','?'))
    acc[key][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[key]+=1
names=sorted({c for v in acc.values() for c in v})
print("%-34s %9s %7s %5s " % ("kernel","grid","lds","calls") + " ".join("%14s"%c[3:17] for c in names))
for k_,v in sorted(acc.items()):
    w=max(cnt[k_],1)
    print("%-34s %9s %7s %5d " % (k_[0],k_[1],k_[2],cnt[k_]) + " ".join("%14.0f"%(v[c]/w) for c in names))
PY
done
rm -rf gpurun_out/pmc_a

# round 2: the attention benchmark + hardware counters of its kernels (two rocprofv3 --pmc passes; counters only with --kernel-trace)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/attn
python tools/bench_attention.py > gpurun_out/attn/attention_utilisation.json 2> gpurun_out/attn/bench.err
cat gpurun_out/attn/attention_utilisation.json
for SET in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"; do
rm -rf gpurun_out/pmc_a
timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d gpurun_out/pmc_a -o c -- python tools/bench_attention.py > gpurun_out/pmc_a.log 2>&1
python - <<'PY' | tee -a gpurun_out/attn/pmc_attention_kernels.txt
import csv, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open('gpurun_out/pmc_a/c_counter_collection.csv')):
    n=r['Kernel_Name']
    if 'attn' not in n and 'gemm_kernel' not in n: continue
    n=re.sub(r'\(anonymous namespace\)::','',n).replace('void toist::','').replace('toist::','')[:40]
    key=(n, r.get('Grid_Size','?'))
    acc[key][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVES': cnt[key]+=1
names=sorted({c for v in acc.values() for c in v})
print("%-40s %9s %5s " % ("kernel","grid","calls") + " ".join("%16s"%c[:16] for c in names))
for k_,v in sorted(acc.items()):
    w=max(cnt[k_],1)
    print("%-40s %9s %5d " % (k_[0],k_[1],cnt[k_]) + " ".join("%16.0f"%(v[c]/w) for c in names))
PY
done
rm -rf gpurun_out/pmc_a gpurun_out/pmc_a.log

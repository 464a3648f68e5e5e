# round 3: hardware counters of the big-tile kernels (two rocprofv3 --pmc passes, counters only with --kernel-trace) over
# tools/r3/conv128.py (3x3 forward / data gradient) and tools/r3/wgrad128.py (grouped weight gradients)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/pmc128; mkdir -p $O; : > $O/pmc_gemm128_kernels.txt
for TOOL in conv128 wgrad128; do
for SET in "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE"; do
rm -rf $O/raw
timeout 900 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $O/raw -o c -- python tools/r3/$TOOL.py > $O/run.log 2>&1
python - "$O" "$TOOL" <<'PY' | tee -a $O/pmc_gemm128_kernels.txt
import csv, collections, re, sys
O, tool = sys.argv[1], sys.argv[2]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(O + '/raw/c_counter_collection.csv')):
    n=r['Kernel_Name']
    if 'gemm128' not in n and 'gemm256' not in n: continue
    n=n.replace('void toist::','').replace('toist::','').replace('(toist_gemm)','')[:34]
    key=(n, r.get('Grid_Size','?'))
    acc[key][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVES': cnt[key]+=1
names=sorted({c for v in acc.values() for c in v})
print("# %s" % tool)
print("%-34s %9s %5s " % ("kernel","grid","calls") + " ".join("%16s"%c[:16] for c in names))
for k_,v in sorted(acc.items()):
    w=max(cnt[k_],1)
    print("%-34s %9s %5d " % (k_[0],k_[1],cnt[k_]) + " ".join("%16.0f"%(v[c]/w) for c in names))
PY
done
done
rm -rf $O/raw $O/run.log

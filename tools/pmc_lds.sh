# LDS behaviour per kernel over two eager training steps (rocprofv3 PMC pass): bank-conflict cycles against LDS-active cycles, LDS instructions, wave cycles
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d gpurun_out/pmc_l -o c -- python bench.py --no-cpu-baseline --no-graph --no-roofline --steps 2 --warmup 1 > gpurun_out/pmc_l.log 2>&1
tail -2 gpurun_out/pmc_l.log | cut -c1-200
python - <<'PY'
import csv, collections, re
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open('gpurun_out/pmc_l/c_counter_collection.csv')):
    n=r['Kernel_Name']
    n=re.sub(r'\(anonymous namespace\)::','',n).replace('void toist::','').replace('toist::','')[:60]
    acc[n][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_WAVE_CYCLES': cnt[n]+=1
rows=sorted(acc.items(), key=lambda kv: -kv[1]['SQ_WAVE_CYCLES'])
print("%-62s %6s %11s %12s %12s %10s %10s %9s" % ("kernel", "calls", "wavecyc(M)", "lds_active(M)", "bank_confl(M)", "confl/act", "LDSinst(M)", "MFMA(M)"))
for n,v in rows[:30]:
    print("%-62s %6d %11.1f %12.2f %12.2f %9.1f%% %10.2f %9.2f" % (n, cnt[n], v['SQ_WAVE_CYCLES']/1e6, v['SQ_LDS_IDX_ACTIVE']/1e6, v['SQ_LDS_BANK_CONFLICT']/1e6,
          100*v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1), v['SQ_INSTS_LDS']/1e6, v['SQ_INSTS_MFMA']/1e6))
PY
rm -rf gpurun_out/pmc_l

# HBM traffic of the fused mask-head stages (rocprofv3 PMC passes as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate
# runs, counters only with --kernel-trace, both in KiB, FETCH_SIZE doubled on gfx950) + LDS conflict / instruction counters, over tools/r5/maskstage_bench.py --iters 3
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmc_ms_$tag -o c -- python tools/r5/maskstage_bench.py --iters 3 > gpurun_out/pmc_ms_$tag.log 2>&1
done
python - <<'PY'
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for d in glob.glob('gpurun_out/pmc_ms_*/'):
    for f in glob.glob(d + '**/c_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'mask_stage_kernel' not in n:
                continue
            key = n[n.index('mask_stage_kernel'):][:48]
            acc[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[key][r['Counter_Name']] += 1
alg = {'<64, 32': 494.8e6, '<32, 16': 989.6e6, '<16, 16': 737.3e6}
print("# per launch; traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB units, gfx950 FETCH correction); algorithmic bytes from tools/r5/maskstage_bench.py")
for k_, v in sorted(acc.items()):
    c = lambda name: v[name] / max(cnt[k_][name], 1)
    a = next((b for p_, b in alg.items() if p_.replace(' ', '') in k_.replace(' ', '')), 0)
    tr = 2 * c('FETCH_SIZE') * 1024 + c('WRITE_SIZE') * 1024
    print(f"{k_:50s} fetch {2 * c('FETCH_SIZE') * 1024 / 1e6:8.1f} MB  write {c('WRITE_SIZE') * 1024 / 1e6:8.1f} MB  traffic {tr / 1e6:8.1f} MB  algorithmic {a / 1e6:7.1f} MB  ratio {tr / a if a else 0:5.2f}")
    print(f"{'':50s} wave cycles {c('SQ_WAVE_CYCLES') / 1e6:8.1f} M  VALU {c('SQ_INSTS_VALU') / 1e6:7.2f} M  MFMA {c('SQ_INSTS_MFMA') / 1e6:6.2f} M  LDS {c('SQ_INSTS_LDS') / 1e6:6.2f} M  "
          f"LDS bank conflict / active {100 * c('SQ_LDS_BANK_CONFLICT') / max(c('SQ_LDS_IDX_ACTIVE'), 1):5.1f} %  waiting {100 * c('SQ_WAIT_ANY') / max(c('SQ_WAVE_CYCLES'), 1):5.1f} % of wave cycles")
PY
rm -rf gpurun_out/pmc_ms_*/

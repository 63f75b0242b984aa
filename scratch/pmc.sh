cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc1 -o p -- python3 scratch/pmc_up.py > gpurun_out/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --output-format csv -d gpurun_out/pmc2 -o p -- python3 scratch/pmc_up.py > gpurun_out/pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ('pmc1','pmc2'):
    f=glob.glob('gpurun_out/%s/*counter_collection.csv'%d)
    if not f: print(d,'no file', glob.glob('gpurun_out/%s/*'%d)); continue
    rows=list(csv.DictReader(open(f[0])))
    agg=collections.OrderedDict()
    for r in rows:
        if 'k_hess_up' not in r['Kernel_Name']: continue
        key=(r['Dispatch_Id'], r['Grid_Size'], r['Workgroup_Size'])
        agg.setdefault(key, {})[r['Counter_Name']] = agg.get(key,{}).get(r['Counter_Name'],0)+float(r['Counter_Value'])
    for k,v in list(agg.items())[-2:]:
        print(d, k, {n:'%.3g'%x for n,x in v.items()})
PY

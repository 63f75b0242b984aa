cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr_l; mkdir -p gpurun_out/tr_l
SMCP_LFSP_V=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_l -o t -- python3 scratch/lfsp_ablate.py > gpurun_out/tr_l/log.txt 2>&1
grep lfsp gpurun_out/tr_l/t_kernel_stats.csv | cut -c1-200
SMCP_LFSP_V=2 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/tr_l2 -o t -- python3 scratch/lfsp_ablate.py > gpurun_out/tr_l/log2.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f=glob.glob('gpurun_out/tr_l2/*counter_collection.csv')
agg=collections.defaultdict(lambda: collections.defaultdict(float)); meta={}
for r in csv.DictReader(open(f[0])):
    nm=r['Kernel_Name'].split('(')[0].replace('void ','').replace('smcp::','')
    if 'lfsp_up' not in nm: continue
    agg[nm][r['Counter_Name']]+=float(r['Counter_Value']); meta[nm]=(r['VGPR_Count'], r['Scratch_Size'])
for nm,c in agg.items(): print(nm, meta[nm], {k: '%.3g'%v for k,v in c.items()})
PY

# usage: bash scratch/trace_kernel.sh <workload> <kernel-substring> : register / scratch use and SQ counters of one kernel
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tr_k; mkdir -p gpurun_out/tr_k
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/tr_k -o t -- python3 bench.py --workload $1 --steps 1 --warmup 0 --no-cpu --no-profile > gpurun_out/tr_k/log.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/tr_k2 -o t -- python3 bench.py --workload $1 --steps 1 --warmup 0 --no-cpu --no-profile > gpurun_out/tr_k/log2.txt 2>&1
python3 - $2 <<'PY'
import csv, glob, collections, sys
for dd in ('tr_k','tr_k2'):
    f=glob.glob('gpurun_out/%s/*counter_collection.csv'%dd)
    if not f: print('missing', dd); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); meta={}; n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        nm=r['Kernel_Name'].split('(')[0].replace('void ','').replace('smcp::','')
        if sys.argv[1] not in nm: continue
        agg[nm][r['Counter_Name']]+=float(r['Counter_Value'])
        meta[nm]=dict(vgpr=r.get('VGPR_Count'), agpr=r.get('Accum_VGPR_Count'), sgpr=r.get('SGPR_Count'), scratch=r.get('Scratch_Size'), lds=r.get('LDS_Block_Size'), grid=r.get('Grid_Size'), wg=r.get('Workgroup_Size'))
    for nm,c in agg.items(): print(nm, meta[nm], dict(c))
PY

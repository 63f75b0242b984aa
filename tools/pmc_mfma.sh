# MFMA utilisation counters of the Gram and family kernels:  bash tools/pmc_mfma.sh  -> gpurun_out/pmc_mfma/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_mfma
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $out/a -o p -- python3 bench.py --no-secondary --steps 2 --no-cpu --no-profile --no-back-solve > $out/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --output-format csv -d $out/b -o p -- python3 bench.py --no-secondary --steps 2 --no-cpu --no-profile --no-back-solve > $out/b.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
for sub in ('a','b'):
    f=glob.glob('%s/%s/*counter_collection.csv'%(out,sub))
    if not f: print('missing',sub, open('%s/%s.log'%(out,sub)).read()[-600:]); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        name=r['Kernel_Name'].split('(')[0].replace('void ','').replace('smcp::','')
        if not any(k in name for k in ('gram_diag','fam_terms','fam_sparse','lf_assemble_lds','lf_assemble_fz','leaf_pairs','lf_up2','mid_chol')): continue
        agg[name][r['Counter_Name']]+=float(r['Counter_Value']); cnt[name].add(r['Dispatch_Id'])
    for n,c in agg.items():
        print(sub, n, 'launches', len(cnt[n]), {k: v/len(cnt[n]) for k,v in c.items()})
PY

# usage: bash scratch/profile_round.sh <tag>   -> gpurun_out/prof_<tag>/
tag=$1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
python3 bench.py --no-secondary --steps 10 > $out/bench.json 2> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-secondary --steps 5 --no-cpu > $out/stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python3 bench.py --no-secondary --steps 2 --no-cpu --no-profile > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python3 bench.py --no-secondary --steps 2 --no-cpu --no-profile > $out/pmc_write.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, json, collections
out=sys.argv[1]
res={}
for which in ('fetch','write'):
    f=glob.glob('%s/pmc_%s/*counter_collection.csv'%(out,which))
    if not f: print('missing', which); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    seen=set()
    for r in csv.DictReader(open(f[0])):
        name=r['Kernel_Name'].split('(')[0].replace('void ','').replace('smcp::','')
        key=(r['Dispatch_Id'])
        agg[name][1]+=float(r['Counter_Value'])
        if key not in seen: seen.add(key); agg[name][0]+=1
    res[which]={k:(v[0],v[1]) for k,v in agg.items()}
json.dump(res, open(out+'/pmc_summary.json','w'), indent=1)
top=sorted(res.get('write',{}).items(), key=lambda kv:-kv[1][1])[:6]
for k,v in top: print(k, 'launches', v[0], 'WRITE_SIZE/launch', v[1]/max(v[0],1), 'FETCH_SIZE/launch', res['fetch'].get(k,(1,0))[1]/max(res['fetch'].get(k,(1,0))[0],1))
PY
cp $out/stats/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
tail -1 $out/bench.json | cut -c1-400

#!/bin/bash
# HBM traffic of the kernels of one bench step from the PMC counters (separate passes, as the MI355X guide prescribes):
# writes gpurun_out/${ROUND}_hbm_traffic.json (copy to profiles/).  FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B
# request of wide coalesced reads), WRITE_SIZE taken as is; both are reported in KB by rocprofv3.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ROUND=${ROUND:-r04}
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o p -- python3 bench.py --no-secondary --steps 2 --warmup 1 --no-cpu --no-profile --no-back-solve > gpurun_out/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o p -- python3 bench.py --no-secondary --steps 2 --warmup 1 --no-cpu --no-profile --no-back-solve > gpurun_out/pmc_w.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, '.')
def collect(d, name):
    f = glob.glob('gpurun_out/%s/*counter_collection.csv' % d)
    out = collections.OrderedDict()
    if not f:
        return out
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name'] != name:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('smcp::', '').replace('(anonymous namespace)::', '')
        e = out.setdefault(k, [0.0, set()])
        e[0] += float(r['Counter_Value'])
        e[1].add(r['Dispatch_Id'])
    return out
F, W = collect('pmc_f', 'FETCH_SIZE'), collect('pmc_w', 'WRITE_SIZE')
ker = {}
for k in sorted(set(F) | set(W)):
    nf = len(F[k][1]) if k in F else 0
    nw = len(W[k][1]) if k in W else 0
    n = max(nf, nw, 1)
    fk = F[k][0] / max(nf, 1) if k in F else 0.0
    wk = W[k][0] / max(nw, 1) if k in W else 0.0
    ker[k] = {"launches": n, "fetch_size_kb_per_launch": round(fk), "write_size_kb_per_launch": round(wk),
              "hbm_bytes_per_launch": int(1024 * (2 * fk + wk))}
top = sorted(ker.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]
json.dump({"_comment": "HBM traffic per launch from rocprofv3 PMC counters, two separate passes (--pmc FETCH_SIZE ; --pmc WRITE_SIZE) of "
           "`python3 bench.py --no-secondary --steps 2 --warmup 1 --no-cpu --no-profile --no-back-solve` on MI355X; FETCH_SIZE doubled per "
           "/opt/skills/guides/MI355X_MICROARCH.md (HBM section), WRITE_SIZE as is; counters in KB",
           "workload": "synth50k nested block-arrow SDP n=50000, 8073 cliques, m=100",
           "csrc_sha256": __import__("bench").csrc_sha256(),
           "kernels": dict(top)}, open('gpurun_out/%s_hbm_traffic.json' % __import__('os').environ.get('ROUND', 'r04'), 'w'), indent=1)
for k, v in top:
    print(k, v)
PY

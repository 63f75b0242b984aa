# Does the speed of k_fam_terms in a process follow address-translation misses?  A few profiled processes: duration and
# TCP_UTCL1_TRANSLATION_MISS / GRBM_UTCL2_BUSY of k_fam_terms.  bash scratch/tlb_mode.sh -> gpurun_out/tlb_mode.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tlb; mkdir -p gpurun_out/tlb
: > gpurun_out/tlb_mode.txt
for i in 1 2 3 4 5; do
  timeout 300 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY --output-format csv -d gpurun_out/tlb/p$i -o t -- python3 bench.py --no-secondary --steps 3 --warmup 1 --no-cpu --no-profile --no-back-solve > gpurun_out/tlb/run$i.log 2>&1
  python3 - <<PY >> gpurun_out/tlb_mode.txt
import csv, glob, collections
f = glob.glob("gpurun_out/tlb/p$i/**/*counter_collection.csv", recursive=True)
k = glob.glob("gpurun_out/tlb/p$i/**/*kernel_trace.csv", recursive=True)
dur = {}
for r in csv.DictReader(open(k[0])):
    if "k_fam_terms" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
acc = collections.defaultdict(dict)
for r in csv.DictReader(open(f[0])):
    if "k_fam_terms" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(acc, key=int):
    print("process $i dispatch", d, "duration %.1f us" % dur.get(d, -1), {k: "%.3g" % v for k, v in acc[d].items()})
PY
done
cat gpurun_out/tlb_mode.txt

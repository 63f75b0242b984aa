# k_fam_terms with and without the placement tuning, alternating fresh processes
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
  for tp in 0 6; do
    SMCP_TIMING=1 timeout -k 10 200 python3 bench.py --no-secondary --steps 10 --warmup 3 --no-cpu --no-back-solve --tune-placement $tp > gpurun_out/pl.json 2> gpurun_out/pl.err
    python3 - <<PY
import json,re
d=json.loads(open("gpurun_out/pl.json").read().strip().splitlines()[-1])
pr=re.findall(r"store-pattern probe ([\d.]+) -> ([\d.]+)", open("gpurun_out/pl.err").read())
print("tune $tp: %.1f solves/s  %.4f ms  k_fam_terms %.3f  assemble %.3f  %s" % (d["value"], d["ms_per_step"], d["kernel_ms_per_step"]["k_fam_terms"], d["kernel_ms_per_step"]["k_lf_assemble_lds_dyn"], pr))
PY
  done
done

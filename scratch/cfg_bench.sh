# quick timing of configs 2 / 3 / 5 without the CPU leg
cd $GRAFT_REPO_ROOT
for w in dense4096 arrow synth50k; do
  timeout 300 python3 bench.py --workload $w --steps 3 --no-cpu > gpurun_out/q_$w.json 2> gpurun_out/q_$w.err || exit 1
done
python3 - <<'PY'
import json
for w in ("dense4096","arrow","synth50k"):
    d=json.loads(open("gpurun_out/q_%s.json"%w).read().strip().splitlines()[-1])
    k=d["kernel_ms_per_step"]
    print(w, round(d["value"],2), round(d["ms_per_step"],2), " ".join("%s=%.2f"%(n.replace("k_",""),v) for n,v in sorted(k.items(), key=lambda x:-x[1])[:9]))
PY

# usage: bash profiles/tools/ab.sh <outdir> NAME:ENV=VAL,ENV=VAL ...   (alternating bench runs, short form; alternating A/B runs of the short bench form)
out=$1; shift; mkdir -p gpurun_out/$out
B="python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 --no-pipeline --no-layout-check --no-side-legs"
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}; envs=${envs//,/ }
  env $envs $B > gpurun_out/$out/${name}_$rep.json 2> gpurun_out/$out/${name}_$rep.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$out/${name}_$rep.json").read().strip().splitlines()[-1]); print("$name", $rep, round(d["value"],1), "lat", round(d["config"].get("single_frame_latency_ms",0),4), flush=True)
except Exception as e: print("$name", "ERR", e)
PY
done; done

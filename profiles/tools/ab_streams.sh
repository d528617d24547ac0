# usage: bash profiles/tools/ab_streams.sh <outdir> NAME:STREAMS:ENV=VAL,... ...
out=$1; shift; mkdir -p gpurun_out/$out
for spec in "$@"; do
  name=${spec%%:*}; rest=${spec#*:}; st=${rest%%:*}; envs=${rest#*:}; envs=${envs//,/ }
  env $envs python bench.py --steps 20 --warmup 5 --no-cpu --no-config5 --no-pipeline --no-layout-check --no-side-legs --streams $st > gpurun_out/$out/${name}.json 2> gpurun_out/$out/${name}.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/$out/${name}.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), "lat", round(d["config"].get("single_frame_latency_ms",0),4), flush=True)
except Exception as e: print("$name", "ERR", e)
PY
done

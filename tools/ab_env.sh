#!/bin/bash
# same-box A/B of an engine switch read from the environment: bench.py twice per setting, interleaved (ABAB)
# usage: bash tools/ab_env.sh <tag> <VAR> <valueA> <valueB> [extra bench args]
T=${1:-abenv}; VAR=$2; A=$3; B=$4; shift 4
mkdir -p gpurun_out/$T
for rep in 1 2; do
  for v in "$A" "$B"; do
    env $VAR=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" > gpurun_out/$T/${VAR}_${v}_$rep.json 2> gpurun_out/$T/${VAR}_${v}_$rep.err || { tail -5 gpurun_out/$T/${VAR}_${v}_$rep.err; exit 1; }
    python3 - <<PY
import json
d = json.loads(open("gpurun_out/$T/${VAR}_${v}_$rep.json").read().strip().splitlines()[-1])
print("$VAR=$v rep $rep:", d["value"], "img/s", d["ms_per_step"], "ms", flush=True)
PY
  done
done

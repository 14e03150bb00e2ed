#!/bin/bash
# same-box A/B of library builds in the whole step: bench.py twice per library, interleaved
# usage: [BENCH_ARGS="--infer --batch 64"] bash tools/ab_lib.sh <tag> <libsuffix1> <libsuffix2> ...   ("-" = the product library)
T=${1:-ablib}; shift
mkdir -p gpurun_out/$T
for rep in 1 2; do
  for v in "$@"; do
    sfx=$v; [ "$v" = "-" ] && sfx=""
    SPG_LIBRARY=$PWD/spegnet_amd/libspegnet_hip$sfx.so timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $BENCH_ARGS > gpurun_out/$T/lib${sfx}_$rep.json 2> gpurun_out/$T/lib${sfx}_$rep.err || { tail -5 gpurun_out/$T/lib${sfx}_$rep.err; exit 1; }
    python3 - <<PY
import json
d = json.loads(open("gpurun_out/$T/lib${sfx}_$rep.json").read().strip().splitlines()[-1])
print("lib$sfx rep $rep:", d["value"], "img/s", d["ms_per_step"], "ms", flush=True)
PY
  done
done

#!/bin/bash
# same-box A/B: product library vs a tagged dev build (SPG_DEV_TAG=_altA SPG_DEV_FLAGS=... python spegnet_amd/build.py --dev ->
# spegnet_amd/libspegnet_hip_dev_altA.so; _lib.load() checks its ABI revision), rocprof per-kernel stats + bench lines
export TMPDIR=/tmp
T=${1:-ab}
mkdir -p gpurun_out/$T
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/pn -o run -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/$T/pn.log 2>&1 || exit 1
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$T/new.json 2> gpurun_out/$T/new.err || exit 1
export SPG_LIBRARY=$PWD/spegnet_amd/libspegnet_hip_dev${SPG_AB_TAG:-_altA}.so
[ -f "$SPG_LIBRARY" ] || { echo "missing $SPG_LIBRARY (build it with SPG_DEV_TAG=${SPG_AB_TAG:-_altA} python spegnet_amd/build.py --dev)"; exit 1; }
timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$T/old.json 2> gpurun_out/$T/old.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/po -o run -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/$T/po.log 2>&1 || exit 1
find gpurun_out/$T -name "*_kernel_trace.csv" -delete
echo ok

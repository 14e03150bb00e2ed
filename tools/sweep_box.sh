#!/bin/bash
export TMPDIR=/tmp
T=$1; shift
mkdir -p gpurun_out/$T
for L in "$@"; do
  SPG_LIBRARY=$PWD/spegnet_amd/$L timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/$L -o run -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/$T/$L.json 2> gpurun_out/$T/$L.err || exit 1
done
find gpurun_out/$T -name "*_kernel_trace.csv" -delete
echo ok

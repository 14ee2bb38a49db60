#!/bin/bash
# A/B of two builds of the library on ONE box: the bench's timed steps alternately under the tree's libdsss.so and under DSSS_LIB=<other.so>.
#   tools/ab_bench.sh <other.so> [rounds=3] [steps=10]
set -euo pipefail
other=${1:?usage: tools/ab_bench.sh <other.so> [rounds] [steps]}; rounds=${2:-3}; steps=${3:-10}
cd "$(cd "$(dirname "$0")/.." && pwd)"
run() { python3 bench.py --cpu-frames 0 --pcie-steps 0 --jobs-in-flight 1 --no-roofline --steps "$steps" --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-10s %.0f frames/s  %.3f ms' % (sys.argv[1], d['value'], d['ms_per_step']))" "$1"; }
for i in $(seq "$rounds"); do run tree; DSSS_LIB="$other" run other; done

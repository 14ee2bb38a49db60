#!/bin/bash
# average duration of the kernels matching a pattern over a short bench.py run, per setting: tools/kstat_bench.sh <pattern> ["ENV=V" ...]
set -euo pipefail
pat=${1:?usage: tools/kstat_bench.sh <pattern> [ENV=V ...]}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)            # the repo this script lives in (GRAFT_REPO_ROOT may be unset outside gpurun)
export TMPDIR=/tmp
cd "$ROOT"
for v in "X=1" "$@"; do
  name=${v%%=*}; val=${v#*=}
  [[ "$v" == *=* && "$name" =~ ^(X|DSSS_[A-Z0-9_]+)$ ]] || { echo "kstat_bench.sh: '$v' is not DSSS_NAME=value" >&2; exit 2; }
  rm -rf gpurun_out/kst
  env "$name=$val" rocprofv3 --kernel-trace --stats -d gpurun_out/kst -o k --output-format csv -- python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-roofline --pcie-steps 0 --jobs-in-flight 1 > gpurun_out/kst.log 2>&1 \
    || { echo "kstat_bench.sh: the profiled bench run failed under $v:" >&2; tail -20 gpurun_out/kst.log >&2; exit 1; }
  [ -s gpurun_out/kst/k_kernel_stats.csv ] || { echo "kstat_bench.sh: no kernel stats under $v" >&2; exit 1; }
  echo "== $v"
  PAT="$pat" python3 - <<'PY'
import csv, os, re
for r in csv.DictReader(open("gpurun_out/kst/k_kernel_stats.csv")):
    if re.search(os.environ["PAT"], r["Name"]):
        print("%-40s calls %5s  avg %9.1f us  total %9.3f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
rm -rf gpurun_out/kst gpurun_out/kst.log

#!/bin/bash
# per-kernel averages of solves of the dumped C3 graph under a kernel trace, one run per setting: tools/pg_kstat_env.sh <pattern> ["DSSS_X=v" ...]
set -euo pipefail
pat=${1:?usage: tools/pg_kstat_env.sh <pattern> [DSSS_NAME=value ...]}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$ROOT"
for v in "X=1" "$@"; do
  name=${v%%=*}; val=${v#*=}
  [[ "$v" == *=* && "$name" =~ ^(X|DSSS_[A-Z0-9_]+)$ ]] || { echo "pg_kstat_env.sh: '$v' is not DSSS_NAME=value" >&2; exit 2; }
  rm -rf gpurun_out/kst
  env "$name=$val" PG_SWEEP_NOPROF=1 rocprofv3 --kernel-trace --stats -d gpurun_out/kst -o k --output-format csv -- python3 tools/pg_sweep.py tools/_data/C3_edges.npz 5 > gpurun_out/kst.log 2>&1 \
    || { echo "pg_kstat_env.sh: the profiled run failed under $v:" >&2; tail -20 gpurun_out/kst.log >&2; exit 1; }
  [ -s gpurun_out/kst/k_kernel_stats.csv ] || { echo "pg_kstat_env.sh: no kernel stats under $v" >&2; exit 1; }
  echo "== $v   $(tail -1 gpurun_out/kst.log | cut -c1-120)"
  PAT="$pat" python3 - <<'PY'
import csv, os, re
for r in csv.DictReader(open("gpurun_out/kst/k_kernel_stats.csv")):
    if re.search(os.environ["PAT"], r["Name"]):
        print("%-40s calls %5s  avg %9.1f us  total %9.3f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
rm -rf gpurun_out/kst gpurun_out/kst.log

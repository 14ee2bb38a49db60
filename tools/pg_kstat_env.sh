#!/bin/bash
# per-kernel averages of solves of the dumped C3 graph under a kernel trace, one run per setting: tools/pg_kstat_env.sh <pattern> ["DSSS_X=v" ...]
set -uo pipefail
pat=${1:?usage: tools/pg_kstat_env.sh <pattern> [DSSS_NAME=value ...]}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd "$ROOT"
for v in "X=1" "$@"; do
  rm -rf gpurun_out/kst
  env "$v" PG_SWEEP_NOPROF=1 rocprofv3 --kernel-trace --stats -d gpurun_out/kst -o k --output-format csv -- python3 tools/pg_sweep.py tools/_data/C3_edges.npz 5 > gpurun_out/kst.log 2>&1 || { echo "run failed under $v"; tail -5 gpurun_out/kst.log; continue; }
  echo "== $v   $(tail -1 gpurun_out/kst.log | cut -c1-120)"
  PAT="$pat" python3 - <<'PY'
import csv, os, re
for r in csv.DictReader(open("gpurun_out/kst/k_kernel_stats.csv")):
    if re.search(os.environ["PAT"], r["Name"]):
        print("%-40s calls %5s  avg %9.1f us  total %9.3f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
rm -rf gpurun_out/kst gpurun_out/kst.log

#!/bin/bash
# average duration of the kernels matching a pattern over a short bench.py run, per setting: tools/kstat_bench.sh <pattern> ["ENV=V" ...]
pat=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in "X=1" "$@"; do
  export "$v"
  rm -rf gpurun_out/kst; rocprofv3 --kernel-trace --stats -d gpurun_out/kst -o k --output-format csv -- python3 bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-roofline > /dev/null 2>&1
  echo "== $v"
  PAT="$pat" python3 - <<'PY'
import csv, os, re
for r in csv.DictReader(open("gpurun_out/kst/k_kernel_stats.csv")):
    if re.search(os.environ["PAT"], r["Name"]):
        print("%-40s calls %5s  avg %9.1f us  total %9.3f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  unset "${v%%=*}"
done
rm -rf gpurun_out/kst

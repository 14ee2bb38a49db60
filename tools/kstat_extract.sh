#!/bin/bash
# average duration of the extraction kernels matching a pattern, for two builds: tools/kstat_extract.sh <pattern> [other.so]
pat=$1; other=$2
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for lib in "$other" ""; do
  if [ -n "$lib" ]; then export DSSS_LIB=$(realpath $lib); else unset DSSS_LIB; fi
  rm -rf gpurun_out/kst; rocprofv3 --kernel-trace --stats -d gpurun_out/kst -o k --output-format csv -- python3 tools/extract_only.py 200 4 > /dev/null 2>&1
  echo "== ${DSSS_LIB:-<tree>}"
  PAT="$pat" python3 - <<'PY'
import csv, os, re
for r in csv.DictReader(open("gpurun_out/kst/k_kernel_stats.csv")):
    if re.search(os.environ["PAT"], r["Name"]):
        print("%-40s calls %5s  avg %9.1f us  total %9.3f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
rm -rf gpurun_out/kst

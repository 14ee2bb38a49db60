#!/bin/bash
# A/B of two builds on one box with the stage breakdown: tools/ab_lc.sh <other.so> [keys...]
other=$1; shift
for i in 1 2; do
  for lib in "$other" ""; do
    if [ -n "$lib" ]; then export DSSS_LIB=$(realpath $lib); else unset DSSS_LIB; fi
    python bench.py --steps 4 --warmup 1 --cpu-frames 0 --pcie-steps 0 --jobs-in-flight 1 2>/dev/null | KEYS="$*" python -c "
import sys,json,os
keys=os.environ.get('KEYS','lc match').split()
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); b=d.get('breakdown_ms',{}); print(os.environ.get('DSSS_LIB','<tree>')[-16:], '%.1f fps %.2f ms'%(d['value'], d['ms_per_step']), {k:b.get(k) for k in keys})
"
  done
done

#!/bin/bash
# A/B of two builds of libdsss.so on the same box: tools/ab_bench.sh <other.so> [bench args]
other=$1; shift
for i in 1 2 3; do
  for lib in "$other" ""; do
    if [ -n "$lib" ]; then export DSSS_LIB=$(realpath $lib); else unset DSSS_LIB; fi
    python bench.py --steps 4 --warmup 1 --cpu-frames 0 --no-roofline "$@" | python -c "
import sys,json,os
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(os.environ.get('DSSS_LIB','<tree>')[-14:], '%.1f fps %.2f ms'%(d['value'], d['ms_per_step']))
"
  done
done

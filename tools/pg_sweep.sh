#!/bin/bash
# A/B sweep of the analysis knobs on a dumped edge set: one process per setting (tools/pg_sweep.py)
G=${1:-gpurun_out/C3_edges.npz}
out=gpurun_out/pg_sweep.log; : > $out
run() { env "$@" python tools/pg_sweep.py $G 7 2>/dev/null | tail -1 >> $out; }
run X=1
for v in 300 1000 1500 2500; do run DSSS_PG_BIN_COST=$v; done
for v in 25 256; do run DSSS_PG_ND_BOTH=$v; done
for v in 16 32; do run DSSS_PG_LEAF=$v; done
cat $out

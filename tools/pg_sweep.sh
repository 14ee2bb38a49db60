#!/bin/bash
# A/B sweep of the analysis knobs on a dumped edge set: one process per setting (tools/pg_sweep.py)
G=${1:-tools/_data/C3_edges.npz}
out=gpurun_out/pg_sweep.log; : > $out
run() { env "$@" PG_SWEEP_NOPROF=1 python tools/pg_sweep.py $G 9 2>/dev/null | tail -1 >> $out; }
run X=1
for v in 150 250 400 900 1500; do run DSSS_PG_BIN_COST=$v; done
for v in 25 256; do run DSSS_PG_ND_BOTH=$v; done
for v in 12 16 32 48; do run DSSS_PG_LEAF=$v; done
run X=2
cat $out

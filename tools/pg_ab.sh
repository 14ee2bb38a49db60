#!/bin/bash
# quick A/B of the pose-graph solve on the dumped C3 graph (tools/dump_graph.py): wall time of dsss_posegraph_solve_edges per setting
G=${1:-tools/_data/C3_edges.npz}
shift
for v in "X=1" "$@"; do env $v python tools/pg_sweep.py $G 9 2>/dev/null | tail -2; done

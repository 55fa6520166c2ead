#!/bin/bash
# Compute part of one rank's term at world = 2, 4, 8 on one GPU (rank 0's landmark shard of the venice shape), with a
# 1-rank RCCL communicator attached so that the kernel sequence is the sharded one; in and out of the hipGraph.
cd "$(dirname "$0")/.." || exit 1
for n in 1 2 4 8; do
  for g in 0 1; do
    echo -n "POVAR_GRAPH_COMM=$g: "
    POVAR_FORCE_COMM=1 POVAR_GRAPH_COMM=$g python3 tools/shard_term_time.py $n ${1:-venice-1778} 2>&1 | grep "world="
  done
  echo -n "p2p push/reduce (in graph): "
  POVAR_FORCE_COMM=1 POVAR_P2P=1 python3 tools/shard_term_time.py $n ${1:-venice-1778} 2>&1 | grep "world="
done

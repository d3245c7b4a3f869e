#!/bin/bash
# nn_large (the search kernel at 16M x 16M) for library variants / chunk sizes of the XCD mapping, same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { python3 profiles/nn_large_only.py 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms_per_search %.4f first %.3f' % (d['ms_per_search'], d['first_search_ms']))"; }
unset ICP_MI355X_LIB; run default
for V in "$@"; do export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_ab_$V.so; run $V; done
export ICP_MI355X_LIB=$PWD/icp_rust_amd/lib/libicp_mi355x_exp.so
for C in 4 64 256 0; do ICP_NN_XCD_CHUNK=$C run "exp chunk=$C"; done
unset ICP_MI355X_LIB; run default

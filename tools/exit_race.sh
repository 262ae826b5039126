#!/bin/bash
# Processes that exit with library background work still in flight (the copy lanes being built, host buffers being released): every exit
# code has to be 0.  Run on the GPU box from the repo root.
cd sparse-linear-algebra_amd || exit 1
run() {   # name, repetitions, python source
    local fails=0
    for i in $(seq 1 "$2"); do python -c "$3" > /dev/null 2>/tmp/exit_race_err.txt || fails=$((fails+1)); done
    echo "exit failures ($1): $fails of $2"
    [ "$fails" -eq 0 ] || tail -3 /tmp/exit_race_err.txt
}
run "context alive at exit" 20 "
import sla_amd as sla
c = sla.Context(0)"
run "context closed right away" 10 "
import sla_amd as sla
c = sla.Context(0); c.close()"
run "matrices then exit" 6 "
import sla_amd as sla, numpy as np, scipy.sparse as sp
c = sla.Context(0)
n = 3000000
A = sp.random(n, n, density=8.0 / n, format='csr', dtype=np.float64, random_state=1)
M = sla.fromCSR((n, n), A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data, ctx=c)
C2 = A.tocoo()
M2 = sla.fromCOO((n, n), C2.row.astype(np.int64), C2.col.astype(np.int64), C2.data, ctx=c)"

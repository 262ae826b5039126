#!/bin/bash
# 40 seeded random banded matrices on 2-5 loopback ranks (one GPU): the ghost-row BiCGSTAB / CGS flows must give the same
# bits (and iteration counts) as the plain sharded flows.  Run on the GPU box: gpurun -- 'bash tools/sweep_ghost_flows.sh'
fail=0
for seed in $(seq 1 40); do
  P=$((2 + seed % 4))
  a=$(timeout 120 python tests/_loopback_worker.py $P fuzz$seed 2>&1 | grep -E "XHASH|LOOPBACK_OK" | tr '\n' ' ')
  b=$(SLA_BICG_GHOST=0 timeout 120 python tests/_loopback_worker.py $P fuzz$seed 2>&1 | grep -E "XHASH|LOOPBACK_OK" | tr '\n' ' ')
  if [ "$a" != "$b" ] || [ -z "$a" ]; then echo "MISMATCH seed $seed P $P: [$a] vs [$b]"; fail=1; fi
done
echo "sweep done fail=$fail"

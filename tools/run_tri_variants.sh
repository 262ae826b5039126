for r in 16384 8192 5440 4096; do
  echo "== block rows $r"
  SLA_TRI_TRACE=/tmp/t.txt timeout 300 python tools/tri_trace.py $r 2>&1 | head -3
  SLA_TRI_TRACE=/tmp/t.txt timeout 300 python tools/tri_trace.py poisson $r 2>&1 | head -2
done

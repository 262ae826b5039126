cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python tools/tile_bench.py 10000000 "DEFAULT=1" "SLA_TILE_SHIFT=18" "SLA_TILE_SHIFT=19" 2>&1 | grep -v "^\[" 
for u in 12 16 23; do echo "== U=$u"; SLA_HIP_LIB=sparse-linear-algebra_amd/lib/libsla_hip_u$u.so timeout 300 python tools/tile_bench.py 10000000 "DEFAULT=1" "SLA_TILE_RELAXED=1" 2>&1 | grep -v "^\[\|^#"; done

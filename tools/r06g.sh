cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_big_nnz.py -m gpu -q -x > $O/pytest_tiles.log 2>&1; echo "rc=$?" >> $O/pytest_tiles.log
timeout 600 python tools/rowown_check.py > $O/rowown_check.txt 2>&1
timeout 900 python tools/tile_bench.py "DEFAULT=1" "SLA_TILES_DEVICE=0" "SLA_TILES_DEVICE=0 SLA_TILE_ROWOWN=1" "SLA_TILE_RELAXED=0" "DEFAULT=1" "SLA_TILES_DEVICE=0 SLA_TILE_ROWOWN=1" > $O/tile_rowown_bench.txt 2>&1
timeout 900 python tools/big_nnz.py > $O/big_nnz_devbuilder.json 2> $O/big_nnz_devbuilder.err
tail -3 $O/pytest_tiles.log; cat $O/rowown_check.txt $O/tile_rowown_bench.txt; tail -5 $O/big_nnz_devbuilder.err

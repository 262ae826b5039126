cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06h; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_loopback.py tests/test_gpu_full_size_oracle.py tests/test_gpu_form_choice.py -m gpu -q > $O/pytest_tiles.log 2>&1; echo "rc=$?" >> $O/pytest_tiles.log
timeout 900 python tools/fuzz_tiles.py 120 606 > $O/fuzz_tiles.txt 2>&1; echo "rc=$?" >> $O/fuzz_tiles.txt
timeout 600 python tools/tile_bench.py "DEFAULT=1" "SLA_TILE_RELAXED=0" "SLA_TILE_RELAXED=0 SLA_TILE_ROWOWN=0" > $O/tile_exact_forms.txt 2>&1
tail -15 $O/pytest_tiles.log; tail -3 $O/fuzz_tiles.txt; cat $O/tile_exact_forms.txt

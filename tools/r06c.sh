cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c; mkdir -p $O
free -g > $O/host.txt; nproc >> $O/host.txt
timeout 900 python -m pytest tests/test_gpu_big_nnz.py tests/test_gpu_tiles.py tests/test_gpu_edge_cases.py tests/test_bcg.py -m gpu -q -x > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log
timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -k "two_ranks or one_json_line" > $O/pytest_bench.log 2>&1; echo "rc=$?" >> $O/pytest_bench.log
timeout 1500 python tools/big_nnz.py --full > $O/big_nnz.json 2> $O/big_nnz.err; echo "rc=$?" >> $O/big_nnz.err
tail -4 $O/pytest_new.log; tail -4 $O/pytest_bench.log; tail -12 $O/big_nnz.err; cat $O/host.txt

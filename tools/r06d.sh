cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_big_nnz.py tests/test_gpu_tiles.py tests/test_gpu_edge_cases.py tests/test_gpu_triangular.py -m gpu -q > $O/pytest_new.log 2>&1; echo "rc=$?" >> $O/pytest_new.log
timeout 1500 python tools/big_nnz.py --full > $O/big_nnz.json 2> $O/big_nnz.err; echo "rc=$?" >> $O/big_nnz.err
timeout 1500 bash tools/csr_ab_lib.sh - abl1 abl2 abl4 abl8 abl16 abl3 abl15 abl31 > $O/wave_ablation.txt 2>&1
tail -4 $O/pytest_new.log; tail -8 $O/big_nnz.err; cat $O/wave_ablation.txt

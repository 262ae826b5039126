cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wave.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -q -x > $O/pytest_wave.log 2>&1; echo "rc=$?" >> $O/pytest_wave.log
timeout 1500 bash tools/csr_ab_lib.sh - wvold wvpre > $O/wave_preops_ab.txt 2>&1
tail -4 $O/pytest_wave.log; cat $O/wave_preops_ab.txt

cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wave.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -q -x > $O/pytest_wave.log 2>&1; echo "rc=$?" >> $O/pytest_wave.log
timeout 600 python -m pytest tests/test_gpu_full_size_oracle.py -m gpu -q -x -k "plain_csr" >> $O/pytest_wave.log 2>&1; echo "rc=$?" >> $O/pytest_wave.log
WAVE_VARIANTS="auto=1:wave_flat=0,auto=1:wave_flat=1" timeout 1200 python tools/wave_ab.py 30 3 > $O/wave_flat_ab.txt 2>&1
grep -E "passed|failed|rc=" $O/pytest_wave.log | tail -4; grep laplace $O/wave_flat_ab.txt

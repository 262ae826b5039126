cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06e; mkdir -p $O
WAVE_VARIANTS="auto=1,auto=1:wave_run=2,auto=1:wave_run=4,auto=1:wave_run=8,auto=1:wave_run=16,auto=1:wave_run=64,wave=408:wave_run=8,wave=604:wave_run=8" timeout 1500 python tools/wave_ab.py 30 2 > $O/wave_run_ab.txt 2>&1
cat $O/wave_run_ab.txt

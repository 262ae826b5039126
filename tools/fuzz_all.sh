cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/fuzz_head; mkdir -p $O
for f in fuzz_formats fuzz_arnoldi fuzz_value_indexed fuzz_tri fuzz_coo fuzz_lds_panels fuzz_tiles fuzz_onchip; do
  echo "== $f" >> $O/fuzz_all.txt
  timeout 600 python tools/$f.py 2>&1 | grep -v "^\[" | tail -3 >> $O/fuzz_all.txt
done
cat $O/fuzz_all.txt | cut -c1-300

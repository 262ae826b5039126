# A/B of the fused Gram-Schmidt step's shapes (build: tools/build_variant_one.sh ao<name> sla_arnoldi_orth.hip -DSLA_AO_CR=.. -DSLA_AO_PIPE=..): GMRES(30) Arnoldi steps / s, 2 M-row banded matrix
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in product aos2 aos3 aot2 aot1 aou3; do
    lib=sparse-linear-algebra_amd/lib/libsla_hip_$v.so; [ $v = product ] && lib=sparse-linear-algebra_amd/lib/libsla_hip.so
    [ -f $lib ] || continue
    echo "$v $(SLA_HIP_LIB=$lib python bench.py --mode gmres --workload banded_2m --steps 120 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-100)"
  done
  echo "launchflow $(SLA_ARN_ORTH=0 python bench.py --mode gmres --workload banded_2m --steps 120 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-100)"
done

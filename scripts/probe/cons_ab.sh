cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for m in 0 1; do
    SABER_HIP_NO_CONSOLIDATE=$m python bench.py --steps 400 --warmup 20 --timed-only "$@" 2>/dev/null | tail -1 | sed "s/^/noconsolidate=$m $r /"
  done
done

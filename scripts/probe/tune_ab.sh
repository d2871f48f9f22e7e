# same-box A/B of the autotuner's timing mode: cold-operand (default) vs back-to-back (SABER_HIP_AUTOTUNE_WARM=1)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for m in 0 1; do
    SABER_HIP_AUTOTUNE_WARM=$m python bench.py --steps 400 --warmup 20 --timed-only "$@" 2>/dev/null | tail -1 | sed "s/^/warm=$m $r /"
  done
done

# if this box is one of the pool's slow ones (forward pass >= 0.27 ms), A/B the end-to-end consolidation pass and the reuse preference on it
cd $GRAFT_REPO_ROOT
python bench.py --steps 300 --warmup 20 --timed-only 2>/dev/null | tail -1 > gpurun_out/boxcheck.json
cat gpurun_out/boxcheck.json
if python -c "import json,sys; sys.exit(0 if json.load(open('gpurun_out/boxcheck.json'))['ms_per_step'] >= 0.27 else 1)"; then
  echo SLOW BOX
  bash scripts/probe/cons_ab.sh | tee gpurun_out/slowbox_cons_ab.txt
  SABER_HIP_AUTOTUNE_WARM=1 python bench.py --steps 400 --warmup 20 --timed-only 2>/dev/null | tail -1 | sed "s/^/warm-tuned /" | tee -a gpurun_out/slowbox_cons_ab.txt
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-b1 --per-op 2>&1 >/dev/null | grep -v "in the chain\|amdgpu" | awk '{print $4}' | sort | uniq -c | sort -rn | tee -a gpurun_out/slowbox_cons_ab.txt
else
  echo fast box
fi

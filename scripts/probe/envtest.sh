cd $GRAFT_REPO_ROOT
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-b1 --tune-cache gpurun_out/env_tune.json > /dev/null 2>&1
run() { echo "== $*"; env "$@" python bench.py --steps 400 --warmup 20 --timed-only --tune-cache gpurun_out/env_tune.json 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['launch_probe'])"; }
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run GPU_FLUSH_ON_EXECUTION=0
run AMD_DIRECT_DISPATCH=1
run A=1

# the round's final evidence run, one box: full GPU test suite, the INT8 profile, every BASELINE configuration, the three other models' profiles, the probes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
(timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/final/pytest.txt
bash scripts/profile_r05.sh r05 > gpurun_out/final/profile_r05.log 2>&1
cp gpurun_out/r05/tune.json profiles/tune.json      # the remaining runs of this script (and the driver's) apply this selection
bash scripts/run_all_configs.sh > gpurun_out/final/configs_summary.txt 2>&1
cp gpurun_out/configs.jsonl gpurun_out/r05/configs.jsonl
bash scripts/profile_model.sh r05_resnet101_int8 --model resnet101 > gpurun_out/final/p101.log 2>&1
bash scripts/profile_model.sh r05_resnet50_fp32 --precision fp32 > gpurun_out/final/pfp32.log 2>&1
bash scripts/profile_model.sh r05_vgg16_fp32 --model vgg16 --precision fp32 > gpurun_out/final/pvgg.log 2>&1
bash scripts/probe/graph_vs_eager.sh gpurun_out/r05 > gpurun_out/r05/graph_vs_eager_summary.txt 2>&1
python scripts/probe/overlap_probe.py > gpurun_out/r05/overlap_probe.txt 2>&1
bash scripts/probe/r05_call9.sh > /dev/null 2>&1; cp gpurun_out/r05g/worker_vs_threads.txt gpurun_out/r05/worker_vs_threads.txt
bash scripts/probe/r05_call11.sh > /dev/null 2>&1; cp gpurun_out/r05h/gemm_fc6.txt gpurun_out/r05/fc_stream.txt
tail -4 gpurun_out/final/pytest.txt; cat gpurun_out/final/configs_summary.txt | tail -12

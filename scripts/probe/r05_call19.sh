mkdir -p gpurun_out/r05o; O=gpurun_out/r05o
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scripts/probe/r05_call18.sh > /dev/null 2>&1; cp gpurun_out/r05n/worker_quiet.txt $O/worker_ready.txt
python bench.py --steps 400 > $O/bench.json 2> $O/bench.err
grep -v "per request" $O/worker_ready.txt | cut -c1-200 | head -12
python -c "
import json; v=json.load(open('$O/bench.json')); r=v['reference_op_list']; print(v['value'], v['ms_per_step'], v['config']['kernel_selection'], v['roofline']['traffic']); [print(k, r[k].get('images_per_s'), r[k].get('median_ms'), r[k].get('max_ms')) for k in ('worker','worker_6_threads','net_threads_1','net_threads_3','worker_pinned_requests','worker_async_prediction')]; print(r['net_prediction']['ms_per_step'], r['net_prediction']['images_per_s'])"

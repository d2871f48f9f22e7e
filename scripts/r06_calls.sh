#!/bin/bash
# Round 6: every GPU call of the round as ONE parameterised runner (round-5 verdict, hygiene: 19 one-off wrappers were tracked).
#   gpurun --timeout N -- 'bash scripts/r06_calls.sh <step> [args]'        outputs under gpurun_out/r06_<step>/
set -u
STEP=${1:?step}; shift || true
O=gpurun_out/r06_$STEP
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
case $STEP in
baseline)   # this round's box: the headline line, the FP32 lines WITH the x86 FP32 baseline (BASELINE configs[0])
  python bench.py --steps 400 --no-cpu-baseline > $O/bench_b8_int8.json 2> $O/bench_b8_int8.err
  python bench.py --precision fp32 --batch 1 --steps 200 --no-b1 --cpu-seconds 30 > $O/fp32_b1.json 2> $O/fp32_b1.err
  python bench.py --precision fp32 --batch 8 --steps 200 --no-b1 --no-cpu-baseline --per-op > $O/fp32_b8.json 2> $O/fp32_b8_per_op.txt
  python bench.py --model vgg16 --precision fp32 --batch 8 --steps 100 --no-b1 --cpu-seconds 20 > $O/vgg16_b8.json 2> $O/vgg16.err
  nproc > $O/host.txt; lscpu | head -20 >> $O/host.txt ;;
compact_ab) # the arena with lifetime aliasing against every edge materialised: the headline pass, batch 1, multi-stream; retunes once (new sources)
  python bench.py --steps 400 --no-cpu-baseline --retune --write-tune-cache > $O/compact1.json 2> $O/compact1.err
  python bench.py --steps 400 --no-cpu-baseline --no-b1 --compact-arena 0 > $O/compact0.json 2> $O/compact0.err
  BENCH_NO_COMPACT=1 python bench.py --steps 100 --no-cpu-baseline --compact-arena 0 > $O/compact0_multi.json 2> $O/compact0_multi.err
  python bench.py --steps 400 --no-cpu-baseline --no-b1 > $O/compact1_again.json 2> $O/compact1_again.err
  cp profiles/tune.json $O/tune.json
  python - <<'PY'
import json
for f in ("compact1","compact0","compact0_multi","compact1_again"):
    try:
        d=json.load(open("gpurun_out/r06_compact_ab/%s.json"%f))
        print(f, d["value"], d["ms_per_step"], d["latency_ms"], d["config"].get("arena_mb"), d["config"].get("arena_mb_every_edge"), d["config"]["kernel_selection"])
        if d.get("batch1"): print("   b1", d["batch1"])
        if d.get("multi_stream"): print("   multi", d["multi_stream"])
        r=d.get("reference_op_list") or {}
        for k in ("net_prediction","worker","worker_6_threads","net_threads_1","net_threads_3"):
            if k in r: print("   ",k,{a:b for a,b in r[k].items() if a!="what"})
    except Exception as e: print(f,"ERR",e)
PY
  ;;
worker_ab)  # Worker<MI355X, INT8> / Net::prediction with the plan's arena compacted (default for shared-device plans) and not
  python - <<'PY'
import os, subprocess, tempfile, json, sys
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
exe = os.path.join("integration", "_build", "test_net_mi355x.bin")
model = W.framework_model(W.build_model("resnet50"), "int8")
scales = W.calibrate(model, W.make_input(2))
out = []
with tempfile.TemporaryDirectory() as td:
    base = W.build_model("resnet50")
    mt, wb = NM.write_model(base, dict(scales), 8, td, "int8", calibrator_config=True)
    W.make_input(8).tofile(os.path.join(td, "input.bin"))
    for rep in range(2):
        for compact in ("0", "1"):
            env = dict(os.environ, SABER_MI355X_NET_PLAN_COMPACT=compact)
            for threads in (3, 6):
                r = subprocess.run([os.path.abspath(exe), mt, wb, os.path.join(td, "input.bin"), td, "worker", str(threads), "900"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, errors="replace", cwd=td, env=env, timeout=600)
                wt = open(os.path.join(td, "worker.txt")).read().split()
                f = {wt[i]: wt[i + 1] for i in range(0, len(wt) - 1, 2)}
                out.append("rep %d compact %s worker threads %d: %s images/s median %s ms max %s ms mismatches %s" % (rep, compact, threads, f["images_per_s"], f["median_ms"], f["max_ms"], f["mismatches"]))
            r = subprocess.run([os.path.abspath(exe), mt, wb, os.path.join(td, "input.bin"), td, "300"], capture_output=True, text=True, cwd=td, env=env, timeout=600)
            tt = open(os.path.join(td, "timing.txt")).read().split()
            pl = open(os.path.join(td, "plan.txt")).read().split("\n")
            out.append("rep %d compact %s Net::prediction %s ms (op loop %s) | %s" % (rep, compact, tt[tt.index("ms_per_prediction") + 1], tt[tt.index("ms_per_prediction_op_loop") + 1], pl[1]))
open("gpurun_out/r06_worker_ab/worker_compact_ab.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
  ;;
oploop)     # where the unplanned operator loop's host time goes; lazy against eager event records; HIP API split under rocprofv3
  python - <<'PY'
import os, subprocess, tempfile, sys, shutil
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
O = "gpurun_out/r06_oploop"
exe = os.path.abspath(os.path.join("integration", "_build", "test_net_mi355x.bin"))
model = W.framework_model(W.build_model("resnet50"), "int8")
scales = W.calibrate(model, W.make_input(2))
td = tempfile.mkdtemp()
mt, wb = NM.write_model(W.build_model("resnet50"), dict(scales), 8, td, "int8", calibrator_config=True)
W.make_input(8).tofile(os.path.join(td, "input.bin"))
lines = []
for rep in range(2):
    for eager in ("1", "0"):
        env = dict(os.environ, SABER_MI355X_EAGER_EVENTS=eager)
        r = subprocess.run([exe, mt, wb, os.path.join(td, "input.bin"), td, "300"], capture_output=True, text=True, cwd=td, env=env, timeout=900)
        tt = open(os.path.join(td, "timing.txt")).read().split()
        lines.append("rep %d eager_events %s: prediction (plan) %s ms, operator loop %s ms" % (rep, eager, tt[tt.index("ms_per_prediction") + 1], tt[tt.index("ms_per_prediction_op_loop") + 1]))
        shutil.copy(os.path.join(td, "op_loop.txt"), os.path.join(O, "op_loop_eager%s_rep%d.txt" % (eager, rep)))
open(os.path.join(O, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
open(os.path.join(O, "cmd.txt"), "w").write(" ".join([exe, mt, wb, os.path.join(td, "input.bin"), td, "100"]) + "\n" + td + "\n")
PY
  CMD=$(head -1 $O/cmd.txt); TD=$(tail -1 $O/cmd.txt)
  (cd $TD && SABER_MI355X_NET_PLAN=0 rocprofv3 --hip-runtime-trace --stats -d $TD/hiptrace -o h -- $CMD > $TD/hiptrace.log 2>&1)
  find $TD/hiptrace -name '*stats*' | head; for f in $(find $TD/hiptrace -name '*hip_api_stats*.csv' -o -name '*_stats.csv' | head -3); do cp $f $O/; done
  ls $O ;;
stage_b1)   # the XCD-resident res5 stage launch against the ops one by one at batch 1 / 2 / 8 (round-5 verdict item 3: re-measure at batch 1)
  for n in 1 2 8; do python scripts/probe/stage_time.py $n; done > $O/stage_time.txt 2>&1; cat $O/stage_time.txt ;;
fp32)       # ResNet50 FP32 b8 / b1 (+ VGG16) bench lines with the per-op table
  python bench.py --precision fp32 --batch 8 --steps 200 --no-b1 --no-cpu-baseline --per-op > $O/fp32_b8.json 2> $O/fp32_b8_per_op.txt
  python bench.py --precision fp32 --batch 1 --steps 200 --no-b1 --no-cpu-baseline --per-op > $O/fp32_b1.json 2> $O/fp32_b1_per_op.txt
  SABER_HIP_STEM_F32_TILE=1 python bench.py --precision fp32 --batch 8 --steps 200 --no-b1 --no-cpu-baseline --per-op > $O/fp32_b8_t1.json 2> $O/fp32_b8_t1_per_op.txt
  SABER_HIP_STEM_F32_TILE=2 python bench.py --precision fp32 --batch 8 --steps 200 --no-b1 --no-cpu-baseline --per-op > $O/fp32_b8_t2.json 2> $O/fp32_b8_t2_per_op.txt
  SABER_HIP_STEM_F32_TILE=3 python bench.py --precision fp32 --batch 8 --steps 200 --no-b1 --no-cpu-baseline --per-op > $O/fp32_b8_t3.json 2> $O/fp32_b8_t3_per_op.txt
  SABER_HIP_STEM_F32_TILE=2 python bench.py --precision fp32 --batch 1 --steps 200 --no-b1 --no-cpu-baseline --per-op > $O/fp32_b1_t2.json 2> $O/fp32_b1_t2_per_op.txt
  python - <<'PY'
import json
for f in ("fp32_b8","fp32_b1","fp32_b8_t1","fp32_b8_t2","fp32_b8_t3","fp32_b1_t2"):
    d=json.load(open("gpurun_out/r06_fp32/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["latency_ms"]["p50"], d["config"]["launches"])
PY
  for f in fp32_b8 fp32_b1 fp32_b8_t1 fp32_b8_t2 fp32_b8_t3 fp32_b1_t2; do echo $f; grep -h "^ *0 " $O/${f}_per_op.txt; done ;;
nosplit)    # what bf16-plane FP32 edges could gain AT MOST: the product library against the probe build whose plane split costs one instruction (wrong results, timing only)
  for b in 8 1; do
    python bench.py --precision fp32 --batch $b --steps 200 --no-b1 --no-cpu-baseline --per-op --retune > $O/fp32_b${b}_product.json 2> $O/fp32_b${b}_product_per_op.txt
    SABER_MI355X_LIB=$PWD/anakin_amd/build_probe/libsaber_mi355x_nosplit.so python bench.py --precision fp32 --batch $b --steps 200 --no-b1 --no-cpu-baseline --per-op --retune > $O/fp32_b${b}_nosplit.json 2> $O/fp32_b${b}_nosplit_per_op.txt
  done
  SABER_MI355X_LIB=$PWD/anakin_amd/build_probe/libsaber_mi355x_nosplit.so python bench.py --model vgg16 --precision fp32 --batch 8 --steps 100 --no-b1 --no-cpu-baseline --retune > $O/vgg16_nosplit.json 2> $O/vgg16_nosplit.err
  python bench.py --model vgg16 --precision fp32 --batch 8 --steps 100 --no-b1 --no-cpu-baseline --retune > $O/vgg16_product.json 2> $O/vgg16_product.err
  python - <<'PY'
import json
for f in ("fp32_b8_product","fp32_b8_nosplit","fp32_b1_product","fp32_b1_nosplit","vgg16_product","vgg16_nosplit"):
    try:
        d=json.load(open("gpurun_out/r06_nosplit/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["config"]["launches"], d["config"]["kernel_selection"])
    except Exception as e: print(f, "ERR", e)
PY
  ;;
fctail)     # the FP32 classifier tail: split-K fc (+ softmax in the launch) against the one-workgroup-per-tile fc + the softmax launch
  for b in 8 1; do
    python bench.py --precision fp32 --batch $b --steps 200 --no-b1 --no-cpu-baseline --per-op > $O/split_b$b.json 2> $O/split_b${b}_per_op.txt
    SABER_HIP_FC_F32_SPLITK=0 python bench.py --precision fp32 --batch $b --steps 200 --no-b1 --no-cpu-baseline --per-op > $O/plain_b$b.json 2> $O/plain_b${b}_per_op.txt
  done
  for f in split_b8 plain_b8 split_b1 plain_b1; do echo $f; python -c "import json;d=json.load(open('$O/$f.json'));print(d['value'],d['ms_per_step'],d['config']['launches'])"; tail -n 4 $O/${f}_per_op.txt; done ;;
final)      # the round's evidence run, one box: GPU test suite, the INT8 profile (tune cache, traces, PMC), every BASELINE configuration, the other models' profiles
  mkdir -p gpurun_out/final
  (timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider 2>&1 | tail -15) > gpurun_out/final/pytest.txt
  bash scripts/profile_r06.sh r06 > gpurun_out/final/profile_r06.log 2>&1
  cp gpurun_out/r06/tune.json profiles/tune.json      # the remaining runs of this script apply this selection
  bash scripts/run_all_configs.sh > gpurun_out/final/configs_summary.txt 2>&1
  cp gpurun_out/configs.jsonl gpurun_out/r06/configs.jsonl
  bash scripts/profile_model.sh r06_resnet101_int8 --model resnet101 > gpurun_out/final/p101.log 2>&1
  bash scripts/profile_model.sh r06_resnet50_fp32 --precision fp32 > gpurun_out/final/pfp32.log 2>&1
  bash scripts/profile_model.sh r06_vgg16_fp32 --model vgg16 --precision fp32 > gpurun_out/final/pvgg.log 2>&1
  tail -4 gpurun_out/final/pytest.txt; tail -12 gpurun_out/final/configs_summary.txt ;;
driver)     # what the driver runs at round end: smoke(), then the bench with its exact flags
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
  ( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2> $O/time.txt; grep real $O/time.txt
  python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_driver/bench_driver.json"))
print(d["metric"], d["value"], d["value_p50"], d["ms_per_step"], d["config"]["kernel_selection"], d["config"]["launches"], d["roofline"]["traffic"], d["roofline"].get("traffic_src_sha"), d["cpu_baseline"]["value"], d["batch1"]["p50_ms"])
PY
  ;;
worker_streams)  # Worker<MI355X, INT8> with the plans' streams from saber_hip_serving_streams (distinct hardware queues) against a fresh stream per plan
  python - <<'PY'
import os, subprocess, tempfile, sys
sys.path.insert(0, os.getcwd())
from anakin_amd import workloads as W
from integration import net_model as NM
exe = os.path.join("integration", "_build", "test_net_mi355x.bin")
model = W.framework_model(W.build_model("resnet50"), "int8")
scales = W.calibrate(model, W.make_input(2))
out = []
with tempfile.TemporaryDirectory() as td:
    base = W.build_model("resnet50")
    mt, wb = NM.write_model(base, dict(scales), 8, td, "int8", calibrator_config=True)
    W.make_input(8).tofile(os.path.join(td, "input.bin"))
    for rep in range(2):
        for serving in ("0", "1"):
            env = dict(os.environ, SABER_MI355X_NET_PLAN_SERVING_STREAMS=serving)
            for threads in (2, 3, 4, 6, 8):
                r = subprocess.run([os.path.abspath(exe), mt, wb, os.path.join(td, "input.bin"), td, "worker", str(threads), "900"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, errors="replace", cwd=td, env=env, timeout=600)
                wt = open(os.path.join(td, "worker.txt")).read().split()
                f = {wt[i]: wt[i + 1] for i in range(0, len(wt) - 1, 2)}
                out.append("rep %d serving_streams %s worker threads %d: %s images/s median %s ms max %s ms mismatches %s" % (rep, serving, threads, f["images_per_s"], f["median_ms"], f["max_ms"], f["mismatches"]))
                print(out[-1], flush=True)
open("gpurun_out/r06_worker_streams/worker_serving_streams_ab.txt", "w").write("\n".join(out) + "\n")
PY
  ;;
streams_trace)   # what every kernel of the pass costs with 0 / 3 other passes in flight: kernel traces of 1 and 4 nets on the serving streams
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  for k in 1 4; do
    rocprofv3 --kernel-trace --stats -d $O/trace$k -o t -- python scripts/probe/multi_stream_profile_target.py $k > $O/target$k.log 2>&1
    N=$(grep "launches per net" $O/target$k.log | awk '{print $4}')
    python scripts/rocprof_summary.py $(find $O/trace$k -name '*_results.db' | head -1) 40 $((N*k*100)) > $O/kernel_trace_summary_${k}_in_flight.txt
    rm -rf $O/trace$k
  done
  head -12 $O/kernel_trace_summary_1_in_flight.txt; head -12 $O/kernel_trace_summary_4_in_flight.txt ;;
streams)    # images/s against the number of independent passes in flight
  python scripts/probe/multi_stream_curve.py > $O/multi_stream_curve.txt 2>&1; grep "^batch" $O/multi_stream_curve.txt
  python scripts/probe/multi_stream_curve.py --pick > $O/multi_stream_curve_pick.txt 2>&1; echo "picked, default queues"; grep "^batch" $O/multi_stream_curve_pick.txt
  GPU_MAX_HW_QUEUES=8 python scripts/probe/multi_stream_curve.py --pick > $O/multi_stream_curve_pick_q8.txt 2>&1; echo "picked, GPU_MAX_HW_QUEUES=8"; grep "^batch" $O/multi_stream_curve_pick_q8.txt
  if [ -n "${STREAMS_ALL:-}" ]; then for q in 2 8 16; do   # is the step down past three (batch 8) / four (batch 4) passes the runtime's four hardware queues?
    GPU_MAX_HW_QUEUES=$q python scripts/probe/multi_stream_curve.py > $O/multi_stream_curve_q$q.txt 2>&1; echo "GPU_MAX_HW_QUEUES=$q"; grep "^batch" $O/multi_stream_curve_q$q.txt
  done; fi ;;
pytest)     # a subset of the GPU tests: bash scripts/r06_calls.sh pytest <pytest args...>
  python -m pytest -x -q "$@" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt ;;
*) echo "unknown step $STEP"; exit 2 ;;
esac

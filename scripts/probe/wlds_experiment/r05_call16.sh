mkdir -p gpurun_out/r05l; O=gpurun_out/r05l
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=12 -p no:cacheprovider -k "weights_through_lds" 2>&1 | tail -40) > $O/pytest.txt
SABER_HIP_AUTOTUNE_LOG=1 timeout 300 python bench.py --steps 200 --precision fp32 --no-cpu-baseline --no-b1 --per-op > $O/bench_r50_fp32.json 2> $O/log_r50_fp32.txt
tail -30 $O/pytest.txt; python -c "
import json; v=json.load(open('$O/bench_r50_fp32.json')); print('r50 fp32', v['value'], v['ms_per_step'], v['config']['launches']); [print('  ', k['kernel'], k['launches'], k['avg_us']) for k in v['roofline']['per_kernel'][:18]]"
python - $O/log_r50_fp32.txt <<'PY'
import re, collections, sys
best = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r'autotune \[(\S+ c\d+ k\d+ \dx\d)\] (\S+)\s+([\d.]+) us', line)
    if not m: continue
    key, name, us = m.group(1), m.group(2), float(m.group(3))
    fam = name if 'wlds' in name else 'other'
    d = best.setdefault(key, {})
    if fam not in d or us < d[fam][0]: d[fam] = (us, name)
for key, d in best.items():
    if len(d) < 2: continue
    o = d.get('other', (0, '-'))
    w = sorted((v[0], n) for n, v in d.items() if n != 'other')[:4]
    print(key, 'incumbent %.2f %s |' % o, ' '.join('%s %.2f' % (n.replace('_f32_bf16x3_64ch', ''), u) for u, n in w))
PY

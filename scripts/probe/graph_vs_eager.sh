#!/bin/bash
# scripts/probe/graph_vs_eager.py under the HIP runtime's graph / kernel-argument switches -> <outdir>/graph_vs_eager.txt
# usage (GPU box): bash scripts/probe/graph_vs_eager.sh gpurun_out/r05
set -u
O=${1:-gpurun_out/r05}
mkdir -p $O
: > $O/graph_vs_eager.txt
python scripts/probe/graph_vs_eager.py >> $O/graph_vs_eager.txt 2>$O/graph_vs_eager.err
for kv in DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=1 DEBUG_HIP_GRAPH_BATCH_SIZE=256 \
          DEBUG_HIP_FORCE_GRAPH_QUEUES=1 DEBUG_HIP_KERNARG_COPY_OPT=0 ROC_USE_FGS_KERNARG=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1; do
  env $kv python scripts/probe/graph_vs_eager.py --quick >> $O/graph_vs_eager.txt 2>>$O/graph_vs_eager.err
done
python - <<PY
import json
for l in open("$O/graph_vs_eager.txt"):
    d = json.loads(l)
    print("%-44s eager %.1f us  lib graph %.1f us  %s %s" % (d["env"] or "default", d["eager_us_per_pass"][0], d["lib_graph_us_per_pass"][0],
          d.get("graph_of_K_passes_us_per_pass", ""), d.get("fit", "")))
PY

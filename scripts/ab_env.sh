#!/bin/bash
# Same-box A/B of environment knobs on the working tree (plus ab/old_tree as the reference).
R=${1:-2}; shift
for r in $(seq 1 $R); do
  (cd ab/old_tree && timeout 300 python bench.py --steps 400 --timed-only "$@" 2>/dev/null | sed "s/^/old     $r /")
  timeout 300 python bench.py --steps 400 --timed-only "$@" 2>/dev/null | sed "s/^/new     $r /"
  SABER_NO_IMG=1 timeout 300 python bench.py --steps 400 --timed-only "$@" 2>/dev/null | sed "s/^/noimg   $r /"
  SABER_NO_MAGIC=1 timeout 300 python bench.py --steps 400 --timed-only "$@" 2>/dev/null | sed "s/^/nomagic $r /"
  SABER_NO_MAGIC=1 SABER_NO_IMG=1 timeout 300 python bench.py --steps 400 --timed-only "$@" 2>/dev/null | sed "s/^/neither $r /"
done

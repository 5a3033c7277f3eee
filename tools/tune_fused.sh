#!/bin/bash
# A/B of the fused-kernel tuning knobs (run on the GPU box): prints kernel_ms per variant
for it in 1 2 4; do for un in 4 8 10; do for nt in 0 1; do
  r=$(NMSA_FUSED_ITERS=$it NMSA_FUSED_UNROLL=$un NMSA_FUSED_NT=$nt python bench.py --no-metrics --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms'], d['ms_per_step'])")
  echo "iters=$it unroll=$un nt=$nt kernel_ms,step_ms = $r"
done; done; done

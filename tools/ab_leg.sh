#!/bin/bash
# Same-box A/B of one bench leg against an older build of the library:
#   tools/ab_leg.sh <leg of tools/profile_leg.py> <old libnmsa_hip.so> [rounds]
# alternates old / current (NMSA_LIB_PATH) and prints every "ms" of the leg's result per run.
leg=$1; old=$2; rounds=${3:-3}
for r in $(seq 1 "$rounds"); do
  for v in old cur; do
    if [ "$v" = old ]; then export NMSA_LIB_PATH=$old; else unset NMSA_LIB_PATH; fi
    python3 tools/profile_leg.py "$leg" 2>/dev/null | tail -1 | python3 -c "
import sys, ast
d = ast.literal_eval(sys.stdin.read())
def walk(p, x):
    if isinstance(x, dict):
        if 'ms' in x: print('  %-44s %.4f ms' % (p, x['ms']), ('frac %.3f' % x['frac']) if 'frac' in x else '')
        for k, v in x.items():
            if isinstance(v, dict): walk(p + '.' + k if p else k, v)
print('$v round $r')
walk('', d)"
  done
done

#!/bin/bash
# ab.sh NAME...: kernel times of bench.py with each tokenizer_amd/lib/variants/libl3hip_NAME.so ("base" = the shipped library)
for v in "$@"; do
  if [ "$v" = base ]; then unset L3HIP_LIB; else export L3HIP_LIB=$PWD/tokenizer_amd/lib/variants/libl3hip_$v.so; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-c5 --no-other-configs --steps 10 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$v', 'ms/step %.3f' % r['ms_per_step'], {k: round(x, 3) for k, x in r['kernel_ms'].items()})
" || exit 1
done

#!/bin/bash
# usage: tools/variants.sh <steps> "<bench args>" lib1.so lib2.so ...   (run on the GPU box)
# environment variables pass through to bench.py
steps=$1; shift
extra=$1; shift
for lib in "$@"; do
  HELICON_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps $steps --warmup 1 --no-cpu-baseline --no-extra-legs $extra 2>/dev/null | python -c "
import sys, json, os
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line); k=d['roofline']['kernels']
    print('$lib', '$extra', 'cand/s=%.0f' % d['value'], ' '.join('%s=%.1fus' % (n, v['avg_us']) for n, v in k.items()), 'batch', d['config']['batch'], d['config']['first_pass'], 'truth', d['argmax']['is_truth'])
" || exit 1
done

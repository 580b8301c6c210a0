#!/bin/bash
# A/B of two library builds on one box: the headline bench line (kernel timers included) alternating between the default
# library and libme_amd_<tag>.so (ME_AMD_LIB_TAG builds of minkowskiengine_amd/build.py).  usage: ab_lib_tags.sh <tag> [rounds]
tag=${1:?tag}; rounds=${2:-3}
for r in $(seq 1 $rounds); do for t in "" $tag; do
  ME_AMD_LIB_TAG=$t timeout 300 python bench.py --extra-workloads off --pmc off --cpu-budget 0 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.readline()); k = d['kernels']
print('${t:-default}', 'ms/step', d['ms_per_step'], 'fwd/dgrad/wgrad us', *[round(k[x]['avg_ms'] * 1e3, 1) for x in ('conv_forward', 'conv_dgrad', 'conv_wgrad')])"
done; done

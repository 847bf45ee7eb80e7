#!/bin/bash
# GPU box: bench.py --config 1 on several A/B builds of the library (tools/ab_variants.py) in one call.
#   usage: tools/ab_bench.sh name1 name2 ...     (libdqp_hip_ab_<name>.so; "main" = the shipped library)
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = main ]; then unset DQP_HIP_LIBRARY; else export DQP_HIP_LIBRARY=$PWD/diff-qp-mpc_amd/csrc/libdqp_hip_ab_$n.so; fi
  python bench.py --config 1 --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']; f=[v for kk,v in k.items() if 'forward_kernel' in kk][0]; b=[v for kk,v in k.items() if 'backward_kernel' in kk][0]
print('%-10s rep $rep  %.3f M QP/s  ms/step %.4f  fwd kernel %.2f us  bwd %.2f us  fast mode %.3f M' % ('$n', d['value']/1e6, d['ms_per_step'], f['avg_ms']*1e3, b['avg_ms']*1e3, d['fast_mode']['value']/1e6))"
done; done

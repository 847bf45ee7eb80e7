#!/bin/bash
# GPU box: the small-batch regime (the reference trains at --bsz 128, deqmpc/train.py:46) -- AL_mpc.MPC calls and the DEQ-MPC
# training step at B = 128, three ways: round 2's host loop around the Newton solves ("old"), the one-call solve
# (dqp_al_mpc_solve, "eager"), and hipGraph replays ("graph"); plus the full-size configs 3-5 replayed as graphs.
for spec in "cartpole2l 5" "pendulum_euler 20" "cartpole1l 20"; do set -- $spec
  for mode in old eager graph; do
    extra=""; [ $mode = graph ] && extra="--graph"
    if [ $mode = old ]; then export DQP_BENCH_AL_PER_SOLVE_CALLS=1; else unset DQP_BENCH_AL_PER_SOLVE_CALLS; fi
    python bench.py --config 3 --robot $1 --T $2 --batch 128 --steps 50 --warmup 5 --no-cpu-baseline $extra > gpurun_out/sb_$1_$mode.json 2> gpurun_out/sb_$1_$mode.err || echo "FAIL $1 $mode"
  done
done
for mode in old eager graph; do
  extra=""; [ $mode = graph ] && extra="--graph"
  if [ $mode = old ]; then export DQP_BENCH_AL_PER_SOLVE_CALLS=1; else unset DQP_BENCH_AL_PER_SOLVE_CALLS; fi
  python bench.py --config 5 --batch 128 --steps 30 --warmup 5 --no-cpu-baseline $extra > gpurun_out/sb_cfg5_$mode.json 2> gpurun_out/sb_cfg5_$mode.err || echo "FAIL cfg5 $mode"
done
unset DQP_BENCH_AL_PER_SOLVE_CALLS
for c in 3 4 5; do
  python bench.py --config $c --graph --no-cpu-baseline > gpurun_out/cfg${c}_graph.json 2> gpurun_out/cfg${c}_graph.err || echo "FAIL cfg$c graph"
done

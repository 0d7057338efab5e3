#!/bin/bash
# Runs on the GPU box: everything profiles/<round>/ is made from, one step after the other (each a rocprofv3 run or a
# plain bench run of its own); stops at the first step that fails.  Results under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
step() { echo "== $*"; "$@" || { echo "FAILED: $*"; exit 1; }; }
step env PROF_DIR=prof_cartpole bash tools/collect_profiles.sh
step env PROF_DIR=prof_pong bash tools/collect_profiles.sh --env pong
step env PROF_DIR=prof_breakout bash tools/collect_profiles.sh --env breakout
step env PROF_DIR=prof_spaceinvaders bash tools/collect_profiles.sh --env spaceinvaders
step bash tools/fp64_profile.sh
step bash tools/rk4_sq.sh
bash tools/acrobot_pmc.sh > gpurun_out/acrobot_exact_sq_counters_device_actions.txt 2>&1 || exit 1
bash tools/acrobot_pmc.sh --policy random > gpurun_out/acrobot_exact_sq_counters_in_kernel_actions.txt 2>&1 || exit 1
echo "== sweep"
timeout -k 10 900 python tools/tools_sweep.py > gpurun_out/sweep_all_workloads.txt 2>&1 || exit 1
echo "== traffic"
: > gpurun_out/traffic_all_workloads_vs_algorithmic.txt
while read -r tag args; do
  bash tools/pmc_traffic_of.sh $tag $args >> gpurun_out/traffic_all_workloads_vs_algorithmic.txt 2>&1 || exit 1
done <<LIST
mountaincar --env mountaincar
mountaincarcontinuous --env mountaincarcontinuous
pendulum --env pendulum
acrobot --env acrobot
acrobot_in_kernel_actions --env acrobot --policy random
frozenlake --env frozenlake --num-envs 67108864
frozenlake_8x8_slippery --env frozenlake --map 8x8 --slippery --num-envs 67108864 --policy random
taxi_stub --env taxi --num-envs 67108864
taxi_full --env taxi --taxi-full --num-envs 67108864 --policy random
blackjack --env blackjack --num-envs 67108864
blackjack_upstream_rules --env blackjack --blackjack-full --num-envs 67108864
cartpole_fp64 --env cartpole --mode fp64
pendulum_fp64 --env pendulum --mode fp64
acrobot_rk4_single_step --env acrobot --rk4 --policy random
LIST
echo ALL DONE

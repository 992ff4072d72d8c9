#!/bin/bash
# The suites whose risk is stream order - the pipelined loader, the group layer, shards over their HBM budget, pairs of
# queries and warm handles - under the interpreter's HIPSIM_ASYNC scheduler, three seeds = its three policies (sim_rt.cpp).
#   bash tools/gfx950sim/async_suites.sh [seeds...] > profiles/r06_sim_async_suites.txt
cd "$(dirname "${BASH_SOURCE[0]}")/../.."
PAR="hbm_budget_are_streamed or larger_than_its_hbm_budget or streamed_shard_answers or compose_with_subsets or pairs_of_queries_one_after or query_file_of_mixed_lengths or longer_than_the_device_driven"
for seed in ${*:-1 2 3}; do
  for suite in "tests/test_gpu_loading.py" "tests/test_gpu_group.py" "tests/test_gpu_parity.py -k \"$PAR\""; do
    t0=$(date +%s)
    out=$(eval HIPSIM_ASYNC=$seed timeout 5400 tools/gfx950sim/run.sh python -m pytest $suite -m gpu -q -p no:cacheprovider 2>&1 | tail -4)
    echo "HIPSIM_ASYNC=$seed (policy $((seed % 3))) $suite: $(echo "$out" | grep -E "passed|failed|error" | tail -1)  [$(( $(date +%s) - t0 )) s]"
    echo "$out" | grep -E "^FAILED|^ERROR" | head -20
  done
done

#!/bin/bash
# The parity soaks a round's final tree is published with (profiles/rNN_soak.txt): run on a GPU box,
#   bash tools/soak_all.sh "<label of the tree>" > gpurun_out/rNN_soak.txt
# Each leg runs only if the one before it passed.
set -o pipefail
echo "# $(date) $1"
echo '## GPC_FUZZ_SEEDS=4000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q'
GPC_FUZZ_SEEDS=4000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2 || exit 1
echo '## GPC_HIP_HASH_TALL=1 GPC_FUZZ_SEEDS=1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q   (40-row hash tiles forced)'
GPC_HIP_HASH_TALL=1 GPC_FUZZ_SEEDS=1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2 || exit 1
echo '## python tools/soak_modes.py 600 0 all'
python tools/soak_modes.py 600 0 all 2>&1 | tail -4 || exit 1
echo '## python -m pytest tests -m gpu -q'
python -m pytest tests -m gpu -q 2>&1 | tail -2 || exit 1

#!/bin/bash
# Round 6, GPU call 11: NeuralNetwork as the policy (sl_policy_net.hip) - its tests, then the whole suite.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r06; mkdir -p $OUT
timeout -k 5 900 python -m pytest tests/test_gpu_network_policy.py -q -m gpu > $OUT/call11_nn.log 2>&1
echo "nn rc=$?"; tail -30 $OUT/call11_nn.log
timeout -k 5 1500 python -m pytest tests -q -m gpu > $OUT/call11_all.log 2>&1
echo "all rc=$?"; tail -4 $OUT/call11_all.log

#!/bin/bash
# Round 5, GPU call 6b: final tree - the whole GPU suite, one bench line per configuration, the
# driver's command (with the CPU baseline), the smoke test.  (The counter passes of call 6 stay
# valid: kernel sources unchanged; profiles/pmc_*.json travel with the tree.)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05_final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/r05_pytest_gpu.log 2>&1
tail -4 $O/r05_pytest_gpu.log
bash tools/bench_configs.sh > $O/configs.txt 2>&1
cp gpurun_out/configs.jsonl $O/r05_configs.jsonl
cut -c1-200 $O/configs.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r05_bench_driver_command.log 2>$O/bench.err
cut -c1-1200 $O/r05_bench_driver_command.log
timeout 600 python bench.py > $O/r05_bench_default.log 2>>$O/bench.err
cut -c1-400 $O/r05_bench_default.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05_smoke.log 2>&1
tail -3 $O/r05_smoke.log

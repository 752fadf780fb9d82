#!/bin/bash
# Round 5, GPU call 6: the evidence run on the final tree - counter passes and kernel stats
# (tools/profile_r05.sh; they write profiles/pmc_traffic.json and profiles/pmc_valu.json, which
# bench.py then reports), one bench line per configuration, the driver's command, the smoke test.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05_final; mkdir -p $O
bash tools/profile_r05.sh > $O/profile.log 2>&1
tail -40 $O/profile.log | cut -c1-180
cp profiles/pmc_traffic.json profiles/pmc_valu.json $O/ 2>/dev/null
bash tools/bench_configs.sh > $O/configs.txt 2>&1
cp gpurun_out/configs.jsonl $O/r05_configs.jsonl
cat $O/configs.txt | cut -c1-200
timeout 900 python bench.py --steps 20 --warmup 5 > $O/r05_bench_driver_command.log 2>$O/bench.err
cut -c1-900 $O/r05_bench_driver_command.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r05_smoke.log 2>&1
tail -3 $O/r05_smoke.log

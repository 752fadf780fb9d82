#!/bin/bash
# summarize kernel resource usage from a verbose build
cd "$(dirname "$0")/.." && python -m safe_learning_amd._build --verbose 2>&1 | grep -E "error|Function Name|VGPRs:|Scratch|Spill|Occupancy" | grep -A5 -E "error|Function Name" | grep -v "^--" | paste - - - - - - | sed -e 's/[^ ]*csrc\/[a-z_.]*:[0-9]*:[0-9]*: remark: *//g' -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' -e 's/  */ /g'

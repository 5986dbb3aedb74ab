#!/bin/bash
# Out-of-bounds sweep of the GPU tests: every test file in its own process with the caching allocator OFF
# (PYTORCH_NO_HIP_MEMORY_CACHING=1: each tensor is its own hipMalloc, so a kernel that reads or writes past the end of one is far
# more likely to leave mapped memory and raise "Memory access fault by GPU" than inside the allocator's 2 MB+ segments).
# (tests/test_graph_gpu.py is left out: graph capture needs the caching allocator; the full-size CPU-port comparisons for time.)
# Usage on the GPU box:  bash tools/oob_sweep.sh [pytest -k expression]   -> gpurun_out/oob_sweep.txt
mkdir -p gpurun_out
out=gpurun_out/oob_sweep.txt
: > $out
for f in $(ls tests/test_*gpu*.py | grep -v "fullsize_parity\|test_dist_gpu\|test_graph_gpu"); do
  start=$(date +%s)
  PYTORCH_NO_HIP_MEMORY_CACHING=1 timeout 900 python -m pytest "$f" -q -m gpu --capture=sys ${1:+-k "$1"} > /tmp/oob_one.log 2>&1
  rc=$?
  echo "$f rc=$rc $(( $(date +%s) - start )) s: $(tail -1 /tmp/oob_one.log | cut -c1-120)" >> $out
  if [ $rc -ne 0 ]; then grep -n "Memory access fault\|Fatal Python error\|^FAILED\|Error" /tmp/oob_one.log | head -8 >> $out; grep -n -A6 "Fatal Python error" /tmp/oob_one.log | grep "File" | head -6 >> $out; fi
done
cat $out

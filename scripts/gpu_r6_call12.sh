#!/bin/bash
# round 6, GPU call 12: product / lab split — the full GPU suite on the product libraries (variant tests on the lab build through the lab_build fixture)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r6c12_gpu_suite.log 2>&1; echo "suite exit $?"
tail -8 gpurun_out/r6c12_gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3

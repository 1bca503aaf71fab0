#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r02s_tests.txt; cat gpurun_out/r02s_tests.txt
timeout 500 python bench.py > gpurun_out/r02s_bench.json 2> gpurun_out/r02s_bench.err; tail -1 gpurun_out/r02s_bench.json | cut -c1-400
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

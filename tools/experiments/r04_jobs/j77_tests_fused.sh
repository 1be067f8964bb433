timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4j77_tests.log 2>&1; tail -2 gpurun_out/r4j77_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1

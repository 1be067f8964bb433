timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4j53_tests.log 2>&1; tail -3 gpurun_out/r4j53_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4j53_smoke.log 2>&1; tail -1 gpurun_out/r4j53_smoke.log

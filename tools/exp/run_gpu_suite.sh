mkdir -p gpurun_out/r05h
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r05h/pytest_all.log 2>&1; echo "rc $?" >> gpurun_out/r05h/pytest_all.log; tail -6 gpurun_out/r05h/pytest_all.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

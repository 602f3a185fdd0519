cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_bench_contract.py -x -q --timeout 280 --timeout-method thread > gpurun_out/r2m_contract.log 2>&1; echo rc=$?; tail -5 gpurun_out/r2m_contract.log
timeout -k 10 300 python -u bench.py --steps 5 --warmup 1 > gpurun_out/r2m_bench.log 2>&1; echo rc=$?; tail -1 gpurun_out/r2m_bench.log

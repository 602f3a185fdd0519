cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_gpu_analytic.py -v -s --timeout 300 --timeout-method thread > gpurun_out/r2w_analytic.log 2>&1; echo rc=$?; grep -E "PASS|FAIL|roughness|Error|assert" gpurun_out/r2w_analytic.log | tail -20

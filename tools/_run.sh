cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_cameras.py -v -s --timeout 300 --timeout-method thread > gpurun_out/r2r_cams.log 2>&1; echo rc=$?; grep -E "PASS|FAIL|Error|assert|differ|scene " gpurun_out/r2r_cams.log | tail -45

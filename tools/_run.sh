cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -u -m pytest tests/test_reference_images.py -v -s --timeout 300 --timeout-method thread > gpurun_out/r2v_refimg.log 2>&1; echo rc=$?; grep -E "PASS|FAIL|correlation|Error" gpurun_out/r2v_refimg.log | tail

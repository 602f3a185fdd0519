cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -u tools/config_runs.py > gpurun_out/r2q_config_runs.txt 2>&1; echo rc=$?; cat gpurun_out/r2q_config_runs.txt

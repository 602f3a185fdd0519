cd $GRAFT_REPO_ROOT
export PTX_RAD_LEGACY=1 PTX_EXCHANGE=1
timeout -k 10 200 python -u tools/wave_profile.py --lib tprof --spp 8 --streams 8 > gpurun_out/r2o_tprof_x.txt 2>&1 && timeout -k 10 200 python -u tools/wave_profile.py --lib wprof --spp 8 --streams 8 > gpurun_out/r2o_wprof_x.txt 2>&1; echo rc=$?; cat gpurun_out/r2o_tprof_x.txt gpurun_out/r2o_wprof_x.txt

cd $GRAFT_REPO_ROOT
export PTX_RAD_LEGACY=1
timeout -k 10 300 python -u tools/wave_profile.py --lib wprof --spp 100 --streams 25 > gpurun_out/r2s_wprof.txt 2>&1; echo rc=$?; cat gpurun_out/r2s_wprof.txt

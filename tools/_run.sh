cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u tools/ab.py --spp 100 --rounds 2 --iters 2 --check 1 --measure 1 base:AB_STREAMS=25 base:AB_STREAMS=25:PTX_COST_ORDER=0 base:AB_STREAMS=0 base:AB_STREAMS=0:PTX_COST_ORDER=0 base:AB_STREAMS=10 base:AB_STREAMS=5 > gpurun_out/r2k_ab.txt 2>&1; rc=$?; echo rc=$rc; cut -c1-420 gpurun_out/r2k_ab.txt

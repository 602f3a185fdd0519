cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u tools/ab.py --spp 100 --rounds 2 --iters 2 --check 0 base:AB_STREAMS=25 e1:AB_STREAMS=25 e2:AB_STREAMS=25 e6:AB_STREAMS=25 e8:AB_STREAMS=25 s12:AB_STREAMS=25 s24:AB_STREAMS=25 s32:AB_STREAMS=25 > gpurun_out/r2p_ab.txt 2>&1; rc=$?; echo rc=$rc; tail -10 gpurun_out/r2p_ab.txt

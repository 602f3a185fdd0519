#!/bin/bash
# FETCH_SIZE calibration run (GPU box): known byte counts of tools/ubench_fetch_calib next to rocprofv3's counters.
#   tools/calib.sh <tag>   -> gpurun_out/calib_<tag>/{ubench.txt, pmc_*/..., fetch_calibration.json}
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/calib_$TAG
BIN=$ROOT/tools/bin/ubench_fetch_calib
rm -rf $OUT && mkdir -p $OUT   # counters of an earlier run must not be summed in
[ -x $BIN ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $BIN $ROOT/tools/ubench_fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 $BIN > $OUT/ubench.txt 2>&1 || exit 1
cat $OUT/ubench.txt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BIN > $OUT/log_fetch.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_rdreq -- $BIN > $OUT/log_rdreq.txt 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_32B_sum --output-format csv -d $OUT/pmc_sizes -- $BIN > $OUT/log_sizes.txt 2>&1 || exit 1
python3 $ROOT/tools/calibrate_fetch.py $OUT/ubench.txt $OUT > $OUT/fetch_calibration.json && cat $OUT/fetch_calibration.json

#!/bin/bash
# SQ counters of the radiance kernel for PTX_QUEUE=0 (pathKernel) and 1 (poolKernel): tools/pmc_ab.sh <tag> <spp> <streams>
TAG=$1; SPP=$2; STREAMS=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for q in 0 1; do
  export PTX_QUEUE=$q
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1_q$q -- python3 $ROOT/tools/render_once.py $SPP $STREAMS > $OUT/sq1_q$q.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/sq2_q$q -- python3 $ROOT/tools/render_once.py $SPP $STREAMS > $OUT/sq2_q$q.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/tcp_q$q -- python3 $ROOT/tools/render_once.py $SPP $STREAMS > $OUT/tcp_q$q.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tcc_q$q -- python3 $ROOT/tools/render_once.py $SPP $STREAMS > $OUT/tcc_q$q.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for q in (0, 1):
    tot = collections.defaultdict(float)
    for f in glob.glob("$OUT/*_q%d/*/*_counter_collection.csv" % q):
        for r in csv.DictReader(open(f)):
            if "pathKernel" in r["Kernel_Name"] or "poolKernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
    print("PTX_QUEUE=%d" % q, {k: "%.4g" % v for k, v in sorted(tot.items())})
PY

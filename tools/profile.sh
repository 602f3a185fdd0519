#!/bin/bash
# Profiling passes for the bench workload (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>/).
#   tools/profile.sh <tag> [spp]
# Pass 0: kernel trace + stats of the default bench command.  Passes 1..: PMC counters, one block set
# per run (rocprofv3 cannot split counters over passes), on a reduced-spp launch of the same kernel.
set -u
TAG=${1:-r01}
SPP=${2:-10}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0, '$ROOT/path-tracer_amd/py'); import ptamd; print(ptamd.build_id())" > $OUT/build_id.txt || exit 1
cd /tmp && export TMPDIR=/tmp
run() { # name, timeout, args...   (PROFILE_ONLY="pass pass ..." runs just those)
  local name=$1; shift; local to=$1; shift
  if [ -n "${PROFILE_ONLY:-}" ] && [[ " $PROFILE_ONLY " != *" $name "* ]]; then return 0; fi
  echo "== $name" | tee -a $OUT/log.txt
  timeout -s KILL $to rocprofv3 "$@" >> $OUT/log.txt 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a $OUT/log.txt
  return $rc
}
if [ -z "${PROFILE_ONLY:-}" ]; then
run trace 400 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-seeding || exit 1
fi
B="python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-reference-seeding --spp $SPP"
run pmc_sq1 300 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq1 -- $B || exit 1
run pmc_sq2 300 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -- $B || exit 1
run pmc_tcc 300 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_tcc -- $B || exit 1
# request sizes at the L2's memory side, and the part of them that went on to DRAM (the rest was served by the Infinity Cache)
run pmc_tcc2 300 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_32B_sum --output-format csv -d $OUT/pmc_tcc2 -- $B || exit 1
run pmc_tcc3 300 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_RDREQ_GMI_32B_sum --output-format csv -d $OUT/pmc_tcc3 -- $B || exit 1
run pmc_fetch 300 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B || exit 1
run pmc_write 300 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B || exit 1
run pmc_tcp 300 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/pmc_tcp -- $B || exit 1
run pmc_grbm 300 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_grbm -- $B || exit 1
echo done | tee -a $OUT/log.txt

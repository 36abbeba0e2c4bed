#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + the PMC passes the roofline numbers come from.
# usage: tools/profile_bench.sh <outdir under gpurun_out> [bench args...]
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="$* --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --no-config-legs"
QUICK="--prewarm-s 0.1 --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 200 --warmup 20 $ARGS > $OUT/bench_kt.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 20 --warmup 5 $QUICK $ARGS > $OUT/bench_pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 $QUICK $ARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 20 --warmup 5 $QUICK $ARGS > $OUT/bench_pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $OUT/pmc_tcc -- python3 bench.py --steps 20 --warmup 5 $QUICK $ARGS > $OUT/bench_pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F64 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 20 --warmup 5 $QUICK $ARGS > $OUT/bench_pmc_sq2.log 2>&1
python3 bench.py --steps 200 --warmup 20 $* > $OUT/bench_plain.log 2>&1
tail -1 $OUT/bench_plain.log | cut -c1-400

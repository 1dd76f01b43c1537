#!/bin/bash
# One GPU call that regenerates the evidence under profiles/ (run on the GPU box via gpurun):
#   1. bench.py (with cpu_baseline)                      -> gpurun_out/prof/bench.json
#   2. rocprofv3 --kernel-trace --stats of bench.py      -> gpurun_out/prof/stats_kernel_stats.csv (+ the bench line under rocprof)
#   3. separate --pmc passes (HBM fetch / write bytes, MFMA + SQ occupancy, L2 hits / misses) as MI355X_MICROARCH.md prescribes
# tools/profile_collect.py then condenses them into profiles/rNN_*.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT && mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done"
# one derived counter per pass where the hardware cannot collect them together (FETCH_SIZE + WRITE_SIZE is refused)
pmc() {  # name, counters...
    local name=$1; shift
    # (one step per launch in the counter passes: EFTB_BENCH_COALESCE=1 -- the counters are then per launch of 128 cosmologies)
    EFTB_BENCH_COALESCE=1 timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT -o $name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/$name.err
    echo "$name done"
}
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
pmc pmc_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE
pmc pmc_l2hit TCC_HIT_sum
pmc pmc_l2miss TCC_MISS_sum
# FETCH_SIZE calibration on a known byte count at 8 and 16 bytes per lane
timeout -k 10 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o calib -- python3 tools/fetch_calib.py > $OUT/calib.out 2> $OUT/calib.err
echo "calib done"
ls $OUT

#!/usr/bin/env bash
# One GPU-box call that refreshes everything profiles/ holds for a round:  bash tools/collect_round.sh <tag>
# (GPU tests, bench line, rocprofv3 kernel stats of the same command, the two PMC passes, training bench line)
set -uo pipefail
tag=${1:-r01_x}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a $out/gpu_tests.log
tail -3 $out/gpu_tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --sustained-seconds 0 --pipelined-streams 0 --spinup-seconds 0 > $out/stats_bench.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --sustained-seconds 0 --pipelined-streams 0 --spinup-seconds 0 > $out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --sustained-seconds 0 --pipelined-streams 0 --spinup-seconds 0 > $out/pmc_write.log 2>&1; echo "pmc write rc=$?"
find $out -name "*.csv" | head -20
# summaries written on the box, from the kernel sources that actually ran (the PMC json carries their hash)
fetch_csv=$(find $out/pmc_fetch -name "*_counter_collection.csv" | head -1)
write_csv=$(find $out/pmc_write -name "*_counter_collection.csv" | head -1)
stats_csv=$(find $out/stats -name "*_kernel_stats.csv" | head -1)
[ -n "$fetch_csv" ] && [ -n "$write_csv" ] && python3 tools/pmc_traffic.py "$fetch_csv" "$write_csv" $out/pmc_traffic.json > $out/pmc_traffic.txt
[ -n "$stats_csv" ] && python3 tools/summarize_prof.py "$stats_csv" $out/stats_bench.log $out/kernel_stats.md
# MFMA utilisation of the network kernels (its own pass: SQ + GRBM counters with --kernel-trace only)
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_mfma -- python3 tools/net_only.py 3 > $out/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
python3 tools/mfma_util.py $out/pmc_mfma $out/pmc_mfma.md > /dev/null 2>&1; echo "mfma_util rc=$?"
# CQT stage: instruction / issue counters next to the traffic of the FETCH / WRITE passes above -> the floors table (tools/cqt_floor.py)
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_cqt_a -- python3 tools/cqt_only.py 3 > $out/pmc_cqt_a.log 2>&1; echo "pmc cqt a rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_cqt_b -- python3 tools/cqt_only.py 3 > $out/pmc_cqt_b.log 2>&1; echo "pmc cqt b rc=$?"
[ -f $out/pmc_traffic.json ] && python3 tools/cqt_floor.py $out/pmc_cqt_a $out/pmc_cqt_b $out/pmc_traffic.json $out/cqt_floor.md > /dev/null 2>$out/cqt_floor.err; echo "cqt_floor rc=$?"
# the bench lines last: the PMC summary of THIS build sits in profiles/ (box-local copy; copy it to the repo's profiles/ afterwards), so
# the line's roofline.traffic is filled from counters taken with the same kernel sources
[ -f $out/pmc_traffic.json ] && cp $out/pmc_traffic.json profiles/${tag}_pmc_traffic.json
timeout -k 10 300 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --train > $out/train_bench.json 2> $out/train_bench.err; echo "train bench rc=$?"
timeout -k 10 300 python3 bench.py --precision f32x3 --no-cpu-baseline --sustained-seconds 0 --pipelined-streams 0 > $out/bench_f32x3.json 2> $out/bench_f32x3.err; echo "f32x3 bench rc=$?"
echo "collect_round done"

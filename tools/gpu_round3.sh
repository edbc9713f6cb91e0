#!/bin/bash
# Round-3 measurement session on the GPU box: bench lines of every config, the N > 1 path with two ranks on one GPU,
# rocprofv3 kernel traces, PMC passes (each counter group in its own run, never with sys/hip/hsa traces).
# usage (through gpurun): bash tools/gpu_round3.sh <tag> [pmc-only]
set -o pipefail
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
rm -rf $O/pmc_* $O/trace_*            # rocprofv3 adds files next to an earlier run's: never mix two builds
cd $R
step() {   # step <name> <seconds> <cmd...>
    local name=$1 secs=$2; shift 2
    timeout -k 10 $secs "$@"
    local rc=$?
    echo "$name rc=$rc"
    if [ $rc -ge 124 ]; then echo "$name timed out / was killed: stopping"; ls $O; exit $rc; fi
    return 0
}
if [ "$2" != "pmc-only" ]; then
step bench_c3 300 bash -c "python bench.py > $O/bench_c3.json 2> $O/bench_c3.err"
step bench_c2 300 bash -c "python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err"
step bench_c4 300 bash -c "python bench.py --workload c4 > $O/bench_c4.json 2> $O/bench_c4.err"
step bench_c5 300 bash -c "python bench.py --workload c5 --steps 5 --warmup 5 > $O/bench_c5.json 2> $O/bench_c5.err"
step bench_c5s 300 bash -c "python bench.py --workload c5 --c5-layout separate --steps 5 --warmup 5 > $O/bench_c5_separate_buffers.json 2> $O/bench_c5_separate_buffers.err"
step bench_c5u 300 bash -c "python bench.py --workload c5 --c5-desc u8 --steps 5 --warmup 5 > $O/bench_c5_u8.json 2> $O/bench_c5_u8.err"
step bench_c5u3 300 bash -c "python bench.py --workload c5 --c5-desc u8 --lanes 3 --c5-host-threads 1 --ransac-wg-ids 0 --steps 5 --warmup 5 > $O/bench_c5_u8_3lanes_1thread_spread_ids.json 2> $O/bench_c5_u8_3lanes.err"
step bench_c3p 300 bash -c "python bench.py --pipeline 1 --no-cpu-baseline --no-large > $O/bench_c3_pipelined.json 2> $O/bench_c3_pipelined.err"
step bench_c3a 300 bash -c "python bench.py --hint auto --no-cpu-baseline --no-large > $O/bench_c3_auto_route.json 2> $O/bench_c3_auto_route.err"
step bench_surf 300 bash -c "python bench.py --kind surf --no-cpu-baseline > $O/bench_c3_surf.json 2> $O/bench_c3_surf.err"
step bench_x1 300 bash -c "python bench.py --exercise-exchange --no-cpu-baseline --no-large > $O/bench_c3_exchange_world1.json 2> $O/bench_c3_exchange_world1.err"
step bench_x1s 300 bash -c "python bench.py --exercise-exchange --pipeline 0 --no-cpu-baseline --no-large > $O/bench_c3_exchange_world1_serial.json 2> $O/bench_c3_exchange_world1_serial.err"
step bench_w2 300 bash -c "python bench.py --gpus 2 --backend gloo --single-device --steps 20 --warmup 5 --no-cpu-baseline --no-large --sustain-seconds 0 > $O/bench_c3_world2_gloo.json 2> $O/bench_c3_world2_gloo.err"
step bench_w2s 300 bash -c "python bench.py --gpus 2 --backend gloo --single-device --workload c4 --scaling strong --steps 5 --warmup 2 --no-cpu-baseline --no-large --sustain-seconds 0 > $O/bench_c4_strong_world2_gloo.json 2> $O/bench_c4_strong_world2_gloo.err"
step fallback 300 bash -c "python tools/fallback_perf.py > $O/fallback_perf.json 2> $O/fallback_perf.err"
step mgpu 300 bash -c "python tools/mgpu_stream_bench.py > $O/mgpu_stream.txt 2>&1"
step flann 300 bash -c "python tools/prof_flann.py > $O/flann.txt 2>&1"
step wide 300 bash -c "python tools/sweep_ratio.py 8192 8192 > $O/sweep_ratio_8192.txt 2>&1"
step forms32k 300 bash -c "python tools/sweep_u8.py 32768 32768 '' '12=2' '12=3' '12=4' '12=5' '12=6' > $O/sweep_u8_forms_32k.txt 2>&1"
step forms16k 300 bash -c "python tools/sweep_u8.py 16384 16384 '' '12=5' '12=6' > $O/sweep_u8_forms_16k.txt 2>&1"
step knnstamps 300 bash -c "for o in '12=2' '12=6'; do echo == options \$o; PM_LIB_PATH=points_matching_amd/build/abl/libpm_knnstamps.so python tools/prof_knn_stamps.py 32768 32768 \$o; done > $O/knn_stamps_32k.txt 2>&1"
step stamps 200 bash -c "for f in 1 2; do echo == PM_OPT_RANSAC_FORM \$f; PM_RANSAC_FORM=\$f PM_LIB_PATH=points_matching_amd/build/abl/libpm_rfstamps.so python tools/prof_ransac_stamps.py; done > $O/ransac_stamps.txt 2>&1"
step h2d 200 bash -c "python tools/h2d_ceiling.py > $O/h2d_ceiling.txt 2>&1"
cd /tmp && export TMPDIR=/tmp
step trace_c3 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --sustain-seconds 0 --headline-only --no-large > $O/trace_c3.log 2>&1
step trace_c3_surf 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3_surf -- python3 $R/bench.py --kind surf --steps 20 --warmup 5 --no-cpu-baseline --no-verify --sustain-seconds 0 --headline-only --no-large > $O/trace_c3_surf.log 2>&1
step trace_c4 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python3 $R/bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-verify --sustain-seconds 0 --no-large > $O/trace_c4.log 2>&1
step trace_l32k 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_l32k -- python3 $R/tools/prof_knn.py 32768 32768 128 10 sift 8 > $O/trace_l32k.log 2>&1
fi
pass() {   # pass <dir> <driver + args> -- <counters...>
    local name=$1 drv=$2; shift 2
    mkdir -p $(dirname $O/$name)
    timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/tools/$drv > $O/$name.log 2>&1      # ($drv may carry arguments: unquoted on purpose)
    local rc=$?
    echo "$name rc=$rc"; tail -1 $O/$name.log
    if [ $rc -ge 124 ]; then echo "timed out: stopping"; exit $rc; fi
}
for d in "knn:prof_knn.py" "knn32k:prof_knn.py 32768 32768 128 4 sift 8" "ransac:prof_ransac.py" "ham:prof_hamming.py"; do
  n=${d%%:*}; drv=${d#*:}
  pass pmc_${n}/sq1 "$drv" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS
  pass pmc_${n}/sq2 "$drv" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE
  pass pmc_${n}/fetch "$drv" FETCH_SIZE
  pass pmc_${n}/write "$drv" WRITE_SIZE
done
ls $O

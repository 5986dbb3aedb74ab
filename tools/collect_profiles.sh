#!/bin/bash
# Evidence run on the GPU box (gpurun): bench lines, rocprofv3 kernel stats and PMC passes for the north-star kernels.
# Usage (from the repo root on the box):  bash tools/collect_profiles.sh <tag> <stage>...     -> gpurun_out/<tag>/
#   bench    headline bench line (parity, roofline rows, cpu baseline) + the other configurations
#   trace    rocprofv3 --kernel-trace --stats of the bench step and of tools/kernel_bench.py
#   pmc      HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA busy of the isolated north-star kernels
#   pmcv     SQ issue / wait counters of the VALU-bound kernels (KNN, FPS) over tools/kernel_bench.py --only knn
#   pmcb     HBM traffic of the set-conv forward / adjoint and the all-pairs lookup over the bench command itself
# PMC counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never together with a
# sys / hip / hsa trace), as MI355X_MICROARCH.md prescribes.  Every rocprofv3 command runs under its own `timeout`; a
# canary (a two-kernel trace) guards each stage so a box whose profiler hangs costs one minute, not the whole call
# (-k: rocprofv3 catches SIGTERM and can then sit in its finalisation for good).
set -u
TAG=${1:-r02}
shift
STAGES="${*:-bench trace pmc pmcv pmcb}"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp

canary() {
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/canary -o c -- \
      python $ROOT/tools/kernel_bench.py --reps 1 --only corr3d > $OUT/canary.log 2>&1
  local rc=$?
  rm -rf /tmp/canary
  [ $rc -eq 0 ] || echo "rocprofv3 canary failed (rc=$rc): profiler stages skipped" | tee -a $OUT/canary.log
  return $rc
}

for stage in $STAGES; do
case $stage in
bench)
  ( cd $ROOT && timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench_line.json 2> $OUT/bench_line.err )
  for cfg in camlipwc kitti eval; do
    ( cd $ROOT && timeout 300 python bench.py --config $cfg --steps 5 --warmup 2 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err )
  done
  ;;
trace)
  canary || continue
  # steady-state kernel statistics of the bench step (trace reduced on the box, the raw trace is not kept)
  # single-lane (CAMLI_OVERLAP=0): with the point branch on its second HIP stream the process stalls under
  # rocprofv3's queue interception on this image (host blocked in a library launch, r02 evidence runs); the bench
  # line printed by this very run (events, same single-lane setting) is kept next to the trace for comparison
  CAMLI_OVERLAP=0 CAMLI_FAULT_DUMP=580 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -o b -- \
      python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-isolated > $OUT/trace_bench.log 2>&1
  python $ROOT/tools/trace_stats.py $OUT/trace_bench/b_kernel_trace.csv --steps 5 --top 90 > $OUT/bench_steady_kernel_stats.csv 2>> $OUT/trace_bench.log
  grep '^{"metric"' $OUT/trace_bench.log > $OUT/bench_line_single_lane_traced.json
  cp $OUT/trace_bench/b_kernel_stats.csv $OUT/bench_whole_process_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/trace_bench
  # isolated kernels: rocprofv3 --kernel-trace --stats of tools/kernel_bench.py (the roofline_rows command)
  CAMLI_KB_KINETO=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_kb -o kb -- \
      python $ROOT/tools/kernel_bench.py --reps 5 --json $OUT/kernel_bench_rows.json > $OUT/kernel_bench.log 2>&1
  cp $OUT/trace_kb/kb_kernel_stats.csv $OUT/kernel_bench_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/trace_kb
  ;;
pmc)
  canary || continue
  RX='fps_|knn_|conv3x3_co2|pointconv_dw_fwd|pointconv_mix|corr2d_fwd|corr2d_bwd|allpairs_lookup|gather_cf|gemm_f32_mfma|gemm_fwd_dma|gemm_gf2|gemm_w128|convcl_kernel|wrw_kernel|segment_row_sum|knn_interp|corr3d_gather|weightnet|pointconv_dw_bwd|pointconv_dw_expand'
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $OUT/pmc_$c -o p -- \
        python $ROOT/tools/kernel_bench.py --reps 2 > /dev/null 2>&1
    python $ROOT/tools/pmc_summary.py $OUT/pmc_$c/p_counter_collection.csv > $OUT/pmc_kernel_bench_$c.txt 2>&1
    rm -rf $OUT/pmc_$c
  done
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace \
      --kernel-include-regex 'gemm_f32_mfma|gemm_fwd_dma|gemm_gf2|gemm_w128|convcl_kernel|wrw_kernel|weightnet' --output-format csv -d $OUT/pmc_mfma -o p -- \
      python $ROOT/tools/kernel_bench.py --reps 2 --only 'a' > /dev/null 2>&1
  # ('a' matches the allpairs case only; the GRU2D convolutions in a pass of their own)
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace \
      --kernel-include-regex 'convcl_kernel|wrw_kernel' --output-format csv -d $OUT/pmc_mfma2 -o p -- \
      python $ROOT/tools/kernel_bench.py --reps 2 --only 'gru2d' > /dev/null 2>&1
  python $ROOT/tools/pmc_summary.py $OUT/pmc_mfma2/p_counter_collection.csv > $OUT/pmc_kernel_bench_mfma_gru2d.txt 2>&1
  rm -rf $OUT/pmc_mfma2
  python $ROOT/tools/pmc_summary.py $OUT/pmc_mfma/p_counter_collection.csv > $OUT/pmc_kernel_bench_mfma.txt 2>&1
  rm -rf $OUT/pmc_mfma
  ;;
pmcv)
  canary || continue
  # issue-side counters of the VALU-bound north-star kernels (KNN, FPS): how much of a wave's life is VALU issue, how much
  # is waiting (SQ counters; their own pass, as every PMC set)
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU \
      --kernel-trace --kernel-include-regex 'knn_|fps_' --output-format csv -d $OUT/pmc_valu -o p -- \
      python $ROOT/tools/kernel_bench.py --reps 2 --only 'knn' > /dev/null 2>&1
  python $ROOT/tools/pmc_summary.py $OUT/pmc_valu/p_counter_collection.csv > $OUT/pmc_kernel_bench_valu.txt 2>&1
  rm -rf $OUT/pmc_valu
  ;;
pmcw)
  canary || continue
  # r6: the Winograd convolution family over tools/kernel_bench.py --only wino (three shapes, forward + both adjoints): HBM
  # traffic (separate FETCH / WRITE passes), matrix-pipe busy, and the traffic per launch of the two entry points
  RXW='wino::|gemm_w128_kernel|wrw_planes'
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "$RXW" --output-format csv -d $OUT/pmcw_$c -o p -- \
        python $ROOT/tools/kernel_bench.py --reps 2 --only wino > /dev/null 2>&1
    python $ROOT/tools/pmc_summary.py $OUT/pmcw_$c/p_counter_collection.csv > $OUT/pmc_wino_$c.txt 2>&1
  done
  python $ROOT/tools/pmc_traffic.py $OUT/pmcw_FETCH_SIZE/p_counter_collection.csv $OUT/pmcw_WRITE_SIZE/p_counter_collection.csv \
      "input_transform_kernel<${CAMLI_WINO_TILE:-4}, true, false>+gemm_w128_kernel+output_transform_kernel" camli_wino_conv3x3 > $OUT/traffic_wino_conv3x3.json 2>&1
  python $ROOT/tools/pmc_traffic.py $OUT/pmcw_FETCH_SIZE/p_counter_collection.csv $OUT/pmcw_WRITE_SIZE/p_counter_collection.csv \
      "wrw_planes_kernel+input_transform_kernel<${CAMLI_WINO_TILE:-4}, true, true>+grad_transform_kernel+wrw_reduce_kernel+bias_grad_kernel" camli_wino_wrw > $OUT/traffic_wino_wrw.json 2>&1
  rm -rf $OUT/pmcw_FETCH_SIZE $OUT/pmcw_WRITE_SIZE
  # the 1-D family (GRU2D's convolutions) over the batch-8 row of tools/kernel_bench.py: four entry points share the input transform
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex 'w1d::|wrw::wrw_kernel' --output-format csv -d $OUT/pmcg_$c -o p -- \
        python $ROOT/tools/kernel_bench.py --reps 2 --only 'gru2d B8' > /dev/null 2>&1
    python $ROOT/tools/pmc_summary.py $OUT/pmcg_$c/p_counter_collection.csv > $OUT/pmc_wino1d_$c.txt 2>&1
  done
  for e in "gru_gates:output_transform_1d_kernel<1>~input_transform_1d_kernel~planes_cl_kernel<8" "gru_blend:output_transform_1d_kernel<2>~input_transform_1d_kernel~planes_cl_kernel<4" \
           "conv:output_transform_1d_kernel<0>~input_transform_1d_kernel~planes_cl_kernel<8" "wrw:grad_transform_1d_kernel~input_transform_1d_kernel~wrw::wrw_kernel~wrw_reduce_1d_kernel"; do
    python $ROOT/tools/pmc_traffic.py $OUT/pmcg_FETCH_SIZE/p_counter_collection.csv $OUT/pmcg_WRITE_SIZE/p_counter_collection.csv \
        "${e#*:}" camli_wino1d_${e%%:*} > $OUT/traffic_wino1d_${e%%:*}.json 2>&1
  done
  rm -rf $OUT/pmcg_FETCH_SIZE $OUT/pmcg_WRITE_SIZE
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace \
      --kernel-include-regex 'gemm_w128_kernel|wrw_planes|planes_cl_kernel|wrw::wrw_kernel' --output-format csv -d $OUT/pmcw_mfma -o p -- \
      python $ROOT/tools/kernel_bench.py --reps 2 --only 'wino|gru2d B8' > /dev/null 2>&1
  python $ROOT/tools/pmc_summary.py $OUT/pmcw_mfma/p_counter_collection.csv > $OUT/pmc_wino_mfma.txt 2>&1
  rm -rf $OUT/pmcw_mfma
  ;;
pmcb)
  canary || continue
  for c in FETCH_SIZE WRITE_SIZE; do
    CAMLI_OVERLAP=0 CAMLI_FAULT_DUMP=580 timeout -k 10 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex 'pointconv_dw_fwd|pointconv_dw_bwd|allpairs_lookup' --output-format csv \
        -d $OUT/pmcb_$c -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated > /dev/null 2>&1
    python $ROOT/tools/pmc_summary.py $OUT/pmcb_$c/p_counter_collection.csv > $OUT/pmc_bench_$c.txt 2>&1
  done
  python $ROOT/tools/pmc_traffic.py $OUT/pmcb_FETCH_SIZE/p_counter_collection.csv $OUT/pmcb_WRITE_SIZE/p_counter_collection.csv \
      pointconv_dw_fwd camli_pointconv_dw_fwd > $OUT/traffic_pointconv_dw_fwd.json 2>&1
  rm -rf $OUT/pmcb_FETCH_SIZE $OUT/pmcb_WRITE_SIZE
  ;;
esac
done
ls -la $OUT

#!/bin/bash
# Round-end measurement pass (runs on the GPU box): rocprofv3 stats + HBM counters of the default bench command, the bench lines of
# every configuration, kernel / operator / training / calibrated-forward benchmarks, per-step launch breakdowns and the PMC passes of
# the two MFMA kernels.  Copy what should be judged from gpurun_out/ into profiles/ (named per round).
# usage: tools/measure_all.sh [tag]      (default tag r05)
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out/$TAG
mkdir -p $O
bash tools/profile_bench.sh $TAG > $O/prof.log 2>&1
cp gpurun_out/prof_$TAG/summary.md $O/rocprof_summary.md 2>/dev/null
cp gpurun_out/prof_$TAG/traffic.json $O/traffic.json 2>/dev/null && cp $O/traffic.json profiles/traffic.json
timeout 600 python bench.py > $O/bench_n100.json 2> $O/bench_n100.err
timeout 300 python bench.py --workload c1 > $O/bench_c1_n36.json 2> $O/bench_c1.err
timeout 600 python bench.py --workload c4 --steps 10 > $O/bench_c4_n256.json 2> $O/bench_c4.err
timeout 300 python bench.py --workload c3 > $O/bench_c3_shared.json 2> $O/bench_c3.err
timeout 300 python bench.py --workload c3 --share-scenes 0 --cpu-sample 0 --stress-preds 0 > $O/bench_c3_unshared.json 2>> $O/bench_c3.err
cp gpurun_out/prof_$TAG/roofline_rocprof_merged.json $O/roofline_rocprof.json 2>/dev/null
timeout 300 python tools/bench_kernels.py > $O/kernel_microbench.jsonl 2> $O/kb.err
# the north-star line with round 3's arithmetic (three bf16 pieces, six products) and on the fp32 pipe, for the A/B table of DESIGN 3.4
DFOL_PAIR_MATH=bf16x3 DFOL_DENSE_MATH=bf16x3 timeout 300 python bench.py --steps 50 --cpu-sample 0 --stress-preds 0 --fresh-batches 0 --streamed 0 --sustain 0 > $O/bench_n100_bf16x3.json 2>> $O/bench_n100.err
DFOL_PAIR_MATH=f32 DFOL_DENSE_MATH=f32 timeout 300 python bench.py --steps 50 --cpu-sample 0 --stress-preds 0 --fresh-batches 0 --streamed 0 --sustain 0 > $O/bench_n100_f32pipe.json 2>> $O/bench_n100.err
timeout 300 python tools/bench_ops.py > $O/ops_throughput.jsonl 2> $O/ops.err
for a in "--objects 36" "--objects 100" "--objects 100 --ragged 10" "--objects 100 --calibrator 1" "--objects 100 --mlp-math bf16" "--objects 100 --graph 0" "--objects 100 --calibrator 1 --graph 0" "--objects 100 --hops ragged" "--objects 100 --hops ragged --mlp-math bf16"; do
  timeout 300 python bench.py --mode train --steps 10 $a >> $O/train_step.jsonl 2>> $O/train.err
done
DFOL_BF16_STORE=0 timeout 300 python bench.py --mode train --steps 10 --objects 100 --mlp-math bf16 >> $O/train_step_bf16_fp32_storage.jsonl 2>> $O/train.err
timeout 200 python tools/lab/bf16_store_lab.py > $O/bf16_storage_kernels.txt 2>&1
DFOL_BENCH_FORCE_PG=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout 300 python bench.py --gpus 1 --steps 10 --mode train --objects 36 --overlap-allreduce 1 > $O/train_rccl_world1.json 2> $O/train_rccl1.err
DFOL_BENCH_FORCE_PG=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29612 timeout 300 python bench.py --gpus 1 --steps 10 --mode train --objects 100 --calibrator 1 > $O/train_rccl_world1_cal.json 2>> $O/train_rccl1.err
DFOL_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 10 > $O/bench_2ranks_one_gpu.json 2> $O/bench_2r.err
DFOL_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 10 --mode train --objects 36 > $O/train_2ranks_one_gpu.json 2> $O/train_2r.err
timeout 300 python tools/bench_calibrated.py > $O/calibrated_forward.txt 2>&1
bash tools/step_breakdown.sh ${TAG}_n100 > $O/step_breakdown_n100.md 2>&1
bash tools/step_breakdown.sh ${TAG}_n36 --objects 36 > $O/step_breakdown_n36.md 2>&1
bash tools/step_breakdown.sh ${TAG}_train_n100 --mode train --objects 100 > $O/step_breakdown_train_n100.md 2>&1
bash tools/step_breakdown.sh ${TAG}_train_bf16_n100 --mode train --objects 100 --mlp-math bf16 > $O/step_breakdown_train_bf16_n100.md 2>&1
# what the bf16 matrix pipe sustains per MFMA shape, and a multiply tick of the pair kernel in isolation
mkdir -p gpurun_out/peak
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/lab/mfma_peak.hip -o gpurun_out/peak/mfma_peak > /dev/null 2>&1 && timeout 120 gpurun_out/peak/mfma_peak > $O/mfma_peak.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/lab/tick_model.hip -o gpurun_out/peak/tick_model > /dev/null 2>&1 && timeout 120 gpurun_out/peak/tick_model > $O/tick_model.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/lab/coissue.hip -o gpurun_out/peak/coissue > /dev/null 2>&1 && timeout 120 gpurun_out/peak/coissue > $O/coissue.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/lab/split_accuracy.hip -o gpurun_out/peak/split_accuracy > /dev/null 2>&1 && timeout 120 gpurun_out/peak/split_accuracy > $O/split_accuracy.txt 2>&1
mkdir -p build && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/lab/transc_accuracy.hip -o build/transc_accuracy > /dev/null 2>&1 && timeout 60 build/transc_accuracy > $O/transcendental_accuracy.txt 2>&1
timeout 300 python tools/accuracy_probe.py --tag $TAG > $O/accuracy_probe_n36.json 2> $O/probe.err
timeout 300 python tools/accuracy_probe.py --tag ${TAG}_n100 --objects 100 --questions 16 > $O/accuracy_probe_n100.json 2>> $O/probe.err
bash tools/pmc_run.sh ${TAG}_mfma "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" tools/bench_kernels.py > $O/pmc_mfma_busy.txt 2>&1
bash tools/pmc_run.sh ${TAG}_insts "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" tools/bench_kernels.py > $O/pmc_insts.txt 2>&1
bash tools/pmc_run.sh ${TAG}_fetch "FETCH_SIZE" tools/bench_kernels.py > $O/pmc_fetch_microbench.txt 2>&1
bash tools/pmc_run.sh ${TAG}_write "WRITE_SIZE" tools/bench_kernels.py > $O/pmc_write_microbench.txt 2>&1
bash tools/train_traffic.sh $TAG > $O/train_traffic.log 2>&1
# round 5: the wide dense kernel against the tiled one (time, and the bytes it fetches per launch), the launches of a train step in order,
# the pair kernel's per-tick trace, the A/B of the train step's fused tail
timeout 200 python tools/lab/time_wide.py > $O/wide_kernel.txt 2>&1
DFOL_DENSE_WIDE=1 bash tools/pmc_run.sh ${TAG}_fetch_wide "FETCH_SIZE" tools/lab/time_wide.py child 25600 > $O/pmc_fetch_wide.txt 2>&1
DFOL_DENSE_WIDE=0 bash tools/pmc_run.sh ${TAG}_fetch_tiled "FETCH_SIZE" tools/lab/time_wide.py child 25600 > $O/pmc_fetch_tiled.txt 2>&1
bash tools/lab/train_launch_order.sh ${TAG}_fp32 --graph 0 > /dev/null 2>&1; cp gpurun_out/order_${TAG}_fp32.txt $O/train_launch_order_fp32.txt 2>/dev/null
bash tools/lab/train_launch_order.sh ${TAG}_bf16 --graph 0 --mlp-math bf16 > /dev/null 2>&1; cp gpurun_out/order_${TAG}_bf16.txt $O/train_launch_order_bf16.txt 2>/dev/null
bash tools/lab/ab_train.sh > $O/train_tail_ab.txt 2>&1
tail -1 $O/bench_n100.json | cut -c1-300
tail -1 $O/bench_c1_n36.json | cut -c1-200
tail -1 $O/bench_c4_n256.json | cut -c1-200

#!/bin/bash
# Copies what tools/measure_all.sh <tag> left under gpurun_out/<tag>/ into profiles/ (tracked), named per round.
# usage: tools/collect_profiles.sh [tag]     (default r03)
TAG=${1:-r03}
O=gpurun_out/$TAG
for f in bench_n100.json bench_c1_n36.json bench_c3_shared.json bench_c3_unshared.json bench_c4_n256.json bench_2ranks_one_gpu.json train_2ranks_one_gpu.json train_rccl_world1.json train_rccl_world1_cal.json; do
  grep '^{' $O/$f | tail -1 > profiles/${TAG}_$f
done
grep '^{' $O/train_step.jsonl > profiles/${TAG}_train_step.jsonl
grep '^{' $O/train_step_bf16_fp32_storage.jsonl > profiles/${TAG}_train_step_bf16_fp32_storage.jsonl
grep -v amdgpu.ids $O/bf16_storage_kernels.txt > profiles/${TAG}_bf16_storage_kernels.txt
grep -v amdgpu.ids $O/calibrated_forward.txt > profiles/${TAG}_calibrated_forward.txt
grep -v amdgpu.ids $O/transcendental_accuracy.txt > profiles/${TAG}_transcendental_accuracy.txt
cp $O/rocprof_summary.md profiles/${TAG}_rocprof_summary.md
cp $O/kernel_microbench.jsonl profiles/${TAG}_kernel_microbench.jsonl
cp $O/ops_throughput.jsonl profiles/${TAG}_ops_throughput.jsonl
for f in pmc_mfma_busy pmc_insts pmc_fetch_microbench pmc_write_microbench mfma_peak tick_model; do cp $O/$f.txt profiles/${TAG}_$f.txt; done
cp $O/accuracy_probe_n36.json profiles/${TAG}_accuracy_probe_n36.json
cp $O/accuracy_probe_n100.json profiles/${TAG}_accuracy_probe_n100.json
for f in step_breakdown_n100 step_breakdown_n36 step_breakdown_train_n100 step_breakdown_train_bf16_n100; do cp $O/$f.md profiles/${TAG}_$f.md; done
cp $O/traffic.json profiles/traffic.json
cp $O/roofline_rocprof.json profiles/roofline_rocprof.json
cp $O/train_traffic.md profiles/${TAG}_train_traffic.md 2>/dev/null
for f in pmc_train_fetch_fp32 pmc_train_write_fp32 pmc_train_fetch_bf16 pmc_train_write_bf16; do cp $O/$f.txt profiles/${TAG}_$f.txt 2>/dev/null; done

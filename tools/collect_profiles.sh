#!/bin/bash
# Copies what tools/measure_all.sh <tag> left under gpurun_out/<tag>/ into profiles/ (tracked), named per round.  A source that is missing
# or empty (a leg of measure_all that failed or timed out) is SKIPPED and reported - it never overwrites committed evidence - and the
# script then exits non-zero.
# usage: tools/collect_profiles.sh [tag]     (default r05)
TAG=${1:-r06}
O=gpurun_out/$TAG
missing=0
take() {          # take <source> <destination> [filter]: copy (or, with "json", keep the last JSON line; "lines": all JSON lines; "clean": drop the amdgpu.ids noise)
  local src=$1 dst=$2 how=${3:-copy} tmp
  if [ ! -s "$src" ]; then echo "collect_profiles: missing or empty: $src (kept $dst as it is)"; missing=$((missing+1)); return; fi
  tmp=$(mktemp)
  case $how in
    json)  grep '^{' "$src" | tail -1 > "$tmp" ;;
    lines) grep '^{' "$src" > "$tmp" ;;
    clean) grep -v amdgpu.ids "$src" > "$tmp" ;;
    *)     cp "$src" "$tmp" ;;
  esac
  if [ -s "$tmp" ]; then mv "$tmp" "$dst"; else echo "collect_profiles: nothing usable in $src (kept $dst as it is)"; missing=$((missing+1)); rm -f "$tmp"; fi
}
for f in bench_n100.json bench_n100_bf16x3.json bench_n100_f32pipe.json bench_c1_n36.json bench_c3_shared.json bench_c3_unshared.json bench_c4_n256.json bench_2ranks_one_gpu.json train_2ranks_one_gpu.json train_rccl_world1.json train_rccl_world1_cal.json; do
  take $O/$f profiles/${TAG}_$f json
done
take $O/train_step.jsonl profiles/${TAG}_train_step.jsonl lines
take $O/train_step_bf16_fp32_storage.jsonl profiles/${TAG}_train_step_bf16_fp32_storage.jsonl lines
for f in bf16_storage_kernels calibrated_forward transcendental_accuracy; do take $O/$f.txt profiles/${TAG}_$f.txt clean; done
take $O/rocprof_summary.md profiles/${TAG}_rocprof_summary.md
take $O/kernel_microbench.jsonl profiles/${TAG}_kernel_microbench.jsonl
take $O/ops_throughput.jsonl profiles/${TAG}_ops_throughput.jsonl
for f in pmc_mfma_busy pmc_insts pmc_fetch_microbench pmc_write_microbench mfma_peak tick_model coissue split_accuracy; do take $O/$f.txt profiles/${TAG}_$f.txt; done
take $O/accuracy_probe_n36.json profiles/${TAG}_accuracy_probe_n36.json
take $O/accuracy_probe_n100.json profiles/${TAG}_accuracy_probe_n100.json
for f in step_breakdown_n100 step_breakdown_n36 step_breakdown_train_n100 step_breakdown_train_bf16_n100; do take $O/$f.md profiles/${TAG}_$f.md; done
take $O/traffic.json profiles/traffic.json
take $O/roofline_rocprof.json profiles/roofline_rocprof.json
take $O/train_traffic.md profiles/${TAG}_train_traffic.md
for f in pmc_train_fetch_fp32 pmc_train_write_fp32 pmc_train_fetch_bf16 pmc_train_write_bf16; do take $O/$f.txt profiles/${TAG}_$f.txt; done
for f in wide_kernel pmc_fetch_wide pmc_fetch_tiled train_launch_order_fp32 train_launch_order_bf16 train_tail_ab; do take $O/$f.txt profiles/${TAG}_$f.txt clean; done
if [ $missing -gt 0 ]; then echo "collect_profiles: $missing source(s) skipped"; exit 1; fi

cd $GRAFT_REPO_ROOT
bash tools/scratch/nt3_lab.sh > /dev/null 2>&1
for v in base noload_nosplit; do
  echo "== $v featurizer"; bash tools/pmc_run.sh lab_$v "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" gpurun_out/lab3/$v 25600 512 2048 1 | grep -A6 "lab_nt3"
  echo "== $v gradx"; bash tools/pmc_run.sh lab_${v}_t "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" gpurun_out/lab3/$v 2534400 256 300 0 | grep -A6 "lab_nt3"
done
echo "== microbench"; bash tools/pmc_run.sh peak "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" gpurun_out/peak/mfma_peak | grep -A5 "k32<4>"

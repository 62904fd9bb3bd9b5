cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/peak
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/scratch/mfma_peak.hip -o gpurun_out/peak/mfma_peak && timeout 120 gpurun_out/peak/mfma_peak | tee gpurun_out/peak/mfma_peak.txt

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/peak
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/scratch/tick_model.hip -o gpurun_out/peak/tick_model 2>&1 | grep -E "error|warning: v" ; timeout 120 gpurun_out/peak/tick_model | tee gpurun_out/peak/tick_model.txt

# Scratch: variants of the registers-only dense kernel, timed on the GPU box (which of loads / splitting / MFMA / stores bounds it)
cd $GRAFT_REPO_ROOT/dfol_vqa_amd/csrc
SRC=dfol_dense_nt3.hip
OUT=$GRAFT_REPO_ROOT/gpurun_out/lab3; mkdir -p $OUT
build() { # name, sed script, extra flags
  sed -e "$2" -e "s/linear_act_nt3_kernel/lab_nt3_kernel/g; s/nt3_pack_kernel/lab_pack_kernel/g" $SRC > $OUT/$1.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $3 -I. $OUT/$1.hip ../../tools/scratch/nt3_lab.cpp -L.. -ldfolvqa -Wl,-rpath,$PWD/.. -o $OUT/$1 2>&1 | grep -E "error" | head -5
}
build trace 's/x/x/' -DDFOL_NT3_TRACE
for a in "2534400 256 300 0" "2534400 300 256 1" "25600 512 2048 1"; do timeout 60 $OUT/trace $a; done 2>&1 | tee $OUT/trace.txt

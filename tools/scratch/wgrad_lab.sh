# Scratch: builds variants of the bf16x3 weight-gradient kernel on the GPU box and times them (which of loads / splitting / MFMA bounds it)
cd $GRAFT_REPO_ROOT/dfol_vqa_amd/csrc
SRC=dfol_dense_wgrad.hip
OUT=$GRAFT_REPO_ROOT/gpurun_out/lab; mkdir -p $OUT
build() { # name, sed script
  sed -e "$2" -e "s/wgrad_tn3_kernel/lab_tn3_kernel/g; s/wgrad_reduce_kernel/lab_reduce_kernel/g; s/wgrad_tn4_kernel/lab_tn4_kernel/g; s/wgrad_tn_kernel/lab_tn_kernel/g" $SRC > $OUT/$1.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. $OUT/$1.hip ../../tools/scratch/wgrad_lab.cpp -L.. -ldfolvqa -Wl,-rpath,$PWD/.. -o $OUT/$1 2>&1 | grep -E "error" | head -5
}
NOSPLIT_A='s/w3_split8(v, ap\[t\]\[0\], ap\[t\]\[1\], ap\[t\]\[2\]);/ap[t][0] = w3_u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}; ap[t][1] = w3_u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}; ap[t][2] = ap[t][0] ^ ap[t][1];/'
NOSPLIT_B='s/w3_split8(v, bp\[u\]\[0\], bp\[u\]\[1\], bp\[u\]\[2\]);/bp[u][0] = w3_u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}; bp[u][1] = w3_u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}; bp[u][2] = bp[u][0] ^ bp[u][1];/'
SAMEROWS='s/(int64_t)s \* 16 \* ld_dy/(int64_t)0 * ld_dy/; s/(int64_t)s \* 16 \* ld_x/(int64_t)0 * ld_x/'
NOBAR='s/__builtin_amdgcn_s_barrier();  *\/\/ the workgroup.*$/;/'
build base 's/x/x/' &
build samerows "$SAMEROWS" &
build nosplit "$NOSPLIT_A; $NOSPLIT_B" &
build nobar "$NOBAR" &
wait
build samerows_nosplit "$SAMEROWS; $NOSPLIT_A; $NOSPLIT_B" &
build f32 's/const bool f32_pipe = .*/const bool f32_pipe = true;/' &
wait
for v in base samerows nosplit nobar samerows_nosplit f32; do timeout 120 $OUT/$v; timeout 60 $OUT/$v 25600 512 2048; done 2>&1 | tee $OUT/results.txt

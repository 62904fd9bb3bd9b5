# Scratch: variants of the LDS-staged bf16x3 dense kernel on the pair layer's tall products (what bounds it?)
cd $GRAFT_REPO_ROOT/dfol_vqa_amd/csrc
SRC=dfol_dense_split.hip
OUT=$GRAFT_REPO_ROOT/gpurun_out/lab4; mkdir -p $OUT
build() { # name, sed script
  sed -e "$2" -e "s/linear_act_split_kernel/lab_split_kernel/g; s/linear_pack_w_split_kernel/lab_packw_kernel/g" $SRC > $OUT/$1.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. $OUT/$1.hip ../../tools/scratch/split_lab.cpp -L.. -ldfolvqa -Wl,-rpath,$PWD/.. -o $OUT/$1 2>&1 | grep -E "error" | head -5
}
# no stores: keep one store per 16 (a data-dependent condition keeps the values alive)
NOSTORE='s/yp\[(int64_t)(i \* 16 + e) \* ldy + j \* 16\] = ls_act<ACT>(acc\[i\]\[j\]\[e\] + bv\[j\]);/{ const float v = ls_act<ACT>(acc[i][j][e] + bv[j]); if (v == 123.456f) yp[(int64_t)(i * 16 + e) * ldy + j * 16] = v; }/'
# X rows of the first row block for every workgroup (L2 hits instead of HBM)
SAMEROWS='s/const int m0 = mb \* LS_BM, n0 = nb \* LS_BN;/const int m0 = 0 * mb * LS_BM, n0 = nb * LS_BN;/'
NOSPLIT='s/ls_split8(k < K ? xa\[S\]\[h\]\[0\] : z, k + 4 < K ? xa\[S\]\[h\]\[1\] : z, ph, pm, pl);/ph = u32x4{__float_as_uint(xa[S][h][0].x), __float_as_uint(xa[S][h][0].y), __float_as_uint(xa[S][h][0].z), __float_as_uint(xa[S][h][0].w)}; pm = u32x4{__float_as_uint(xa[S][h][1].x), __float_as_uint(xa[S][h][1].y), __float_as_uint(xa[S][h][1].z), __float_as_uint(xa[S][h][1].w)}; pl = ph ^ pm;/'
build base 's/x/x/' &
build nostore "$NOSTORE" &
build samerows "$SAMEROWS" &
build nosplit "$NOSPLIT" &
wait
build nostore_samerows "$NOSTORE; $SAMEROWS" &
build all "$NOSTORE; $SAMEROWS; $NOSPLIT" &
wait
for v in base nostore samerows nosplit nostore_samerows all; do timeout 120 $OUT/$v; timeout 120 $OUT/$v 2534400 256 300 0; done 2>&1 | tee $OUT/results.txt

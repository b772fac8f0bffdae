# developer tool: second build of the library with in-kernel phase stamps compiled in -> yolo_amd/lib/libty1_dev.so
set -e
cd "$(dirname "$0")/../yolo_amd/csrc"
O=/tmp/ty1_dev_obj; mkdir -p $O
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DTY1_DEV_STAMPS"
for f in common conv_gemm conv_halo dense_gemm train; do /opt/rocm/bin/hipcc $F -c $f.hip -o $O/$f.o & done
/opt/rocm/bin/hipcc $F -fno-honor-nans -c conv_fused.hip -o $O/conv_fused.o &
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -c conv_direct.hip -o $O/conv_direct.o &
for f in decode_nms loss preprocess head; do /opt/rocm/bin/hipcc $F -ffp-contract=off -c $f.hip -o $O/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libty1_dev.so $O/*.o

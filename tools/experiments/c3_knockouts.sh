# developer experiment (one GPU call): where the first-layer kernel's time goes.
#   libty1_dev.so          tools/build_dev.sh; TY1_C3_DBG: 1 drops the output stores, 2 the halo fetch (same instruction stream)
#   libty1_ko{1,2,3}.so    the same build with conv_direct.hip compiled -DC3_KO=1 (no epilogue), 2 (no LDS reads), 3 (both):
#     for ko in 1 2 3; do mkdir -p /tmp/ko$ko; cp /tmp/ty1_dev_obj/*.o /tmp/ko$ko/
#       hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -DTY1_DEV_STAMPS -DC3_KO=$ko -mllvm -amdgpu-mfma-vgpr-form=1 \
#             -c yolo_amd/csrc/conv_direct.hip -o /tmp/ko$ko/conv_direct.o
#       hipcc --offload-arch=gfx950 -shared -fPIC -o yolo_amd/lib/libty1_ko$ko.so /tmp/ko$ko/*.o; done
# Round-2 result at batch 64 (us): shipping stream 108.8; no stores 104.7; no fetch 104.3; neither 102; neither and no
# epilogue 92.4; MFMA floor 76 at 2.3 GHz.
set -e
for d in 0 1 2 3; do
  echo "TY1_C3_DBG=$d"; TY1_LIB=$PWD/yolo_amd/lib/libty1_dev.so TY1_C3_DBG=$d timeout -k 10 100 python tools/layer_bench.py --batch 64 --layers 0 --iters 1000
done
for v in ko1 ko2 ko3; do
  [ -f yolo_amd/lib/libty1_$v.so ] || continue
  echo "lib $v"; TY1_LIB=$PWD/yolo_amd/lib/libty1_$v.so TY1_C3_DBG=3 timeout -k 10 100 python tools/layer_bench.py --batch 64 --layers 0 --iters 1000
done

# developer tool: builds of libty1.so that differ only in conv_fused.hip's compile-time switches, for an A/B inside one GPU call.
#   build (CPU box):  tools/fused_variants.sh build tagA "-DF_KO=1" tagB "-DF_MIN_WAVES=2" ...
#   run   (GPU box):  tools/fused_variants.sh run tagA tagB ...      (each: tools/fused_bench.py with TY1_LIB set)
set -e
cd "$(dirname "$0")/.."
mode=$1; shift
if [ "$mode" = build ]; then
  while [ $# -gt 0 ]; do
    tag=$1; flags=$2; shift 2
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w -fno-honor-nans $flags -c yolo_amd/csrc/conv_fused.hip -o /tmp/conv_fused_$tag.o
    objs=$(ls yolo_amd/csrc/*.o | grep -v conv_fused.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o yolo_amd/lib/libty1_fv_$tag.so $objs /tmp/conv_fused_$tag.o
    echo "built libty1_fv_$tag.so ($flags)"
  done
else
  for tag in "$@"; do
    lib=$PWD/yolo_amd/lib/libty1_fv_$tag.so
    [ "$tag" = base ] && lib=$PWD/yolo_amd/lib/libty1.so
    echo "== $tag: $(TY1_LIB=$lib timeout -k 5 90 python tools/fused_bench.py --rounds 3 2>/dev/null)" || exit 1
  done
fi

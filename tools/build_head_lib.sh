#!/bin/bash
# developer tool (CPU box): build the library from the csrc of a git revision (default HEAD) -> yolo_amd/lib/libty1_head.so,
# for A/Bs of a kernel change against the OLD SOURCE inside one GPU call (TY1_LIB=$PWD/yolo_amd/lib/libty1_head.so)
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
rm -rf /tmp/ty1_head && mkdir -p /tmp/ty1_head
git archive $REV yolo_amd/csrc include | tar -x -C /tmp/ty1_head
mkdir -p /tmp/ty1_head/yolo_amd/lib
make -s -C /tmp/ty1_head/yolo_amd/csrc 2>&1 | grep -E "error|Error" && exit 1
cp /tmp/ty1_head/yolo_amd/lib/libty1.so yolo_amd/lib/libty1_head.so
ls -la yolo_amd/lib/libty1_head.so

#!/bin/bash
# developer tool (CPU box): rebuild the shipping library, then run a command on the GPU box.  usage: tools/gr.sh SECONDS 'command'
set -e
cd "$(dirname "$0")/.."
make -s -C yolo_amd/csrc 2>&1 | grep -E "error|Error" && exit 1
t=$1; shift
/usr/local/graft/bin/gpurun --timeout $t -- "$@" 2>&1 | grep -v "^\[gpurun\] every\|^\[gpurun\] merged"

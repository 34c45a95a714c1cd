#!/bin/bash
# Dev helper: rebuild the HIP library, then run a command on the GPU box.  usage: tools/gpu.sh <timeout> '<cmd>'
set -e
cd "$(dirname "$0")/.."
python smart_tree_amd/build_ext.py > /tmp/build_ext.log 2>&1 || { tail -30 /tmp/build_ext.log; exit 1; }
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2"

#!/bin/bash
# tools/mkvar.sh NAME [extra hipcc flags]  -> build/libmbpe_NAME.so  (a variant library for A/B runs: MBPE_LIB=build/libmbpe_NAME.so)
set -e
cd "$(dirname "$0")/.." && mkdir -p build
n=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden "$@" \
  -Iinclude -Iminbpe-cc_amd/csrc -Iminbpe-cc_amd/host \
  minbpe-cc_amd/csrc/kernels.hip minbpe-cc_amd/csrc/train.cpp minbpe-cc_amd/csrc/encode.hip minbpe-cc_amd/csrc/wide.hip minbpe-cc_amd/host/presplit.cpp minbpe-cc_amd/host/errors.cpp \
  minbpe-cc_amd/host/tokenizer.cpp -ldl -o build/libmbpe_$n.so

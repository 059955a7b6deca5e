#!/bin/bash
# builds gemm4w_probe[_TAG] (device assembly kept in _tmp/TAG/ for inspection): ./build4w.sh [TAG [-D flags...]]
TAG=${1:-}; shift
cd "$(dirname "$0")" && mkdir -p _tmp/x$TAG && cd _tmp/x$TAG && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-unused-value -save-temps "$@" -o ../../gemm4w_probe${TAG:+_$TAG} ../../gemm4w_probe.hip

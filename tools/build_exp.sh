#!/bin/bash
# Build the timestamped copies of the product kernels (tools/gen_exp_*.py -> tools/exp_*.hip -> tools/libexp*.so) with
# the product's flags:  bash tools/build_exp.sh ztime [-DEXP_...]   (ztime -> libexpt.so, btime -> libexpb.so,
# d2m -> libexpd.so, mesh -> libexpm.so)
set -e
cd "$(dirname "$0")/.."
which=$1; shift
case $which in ztime) so=libexpt.so;; btime) so=libexpb.so;; d2m) so=libexpd.so;; mesh) so=libexpm.so;; *) echo "ztime|btime|d2m|mesh"; exit 1;; esac
python tools/gen_exp_$which.py
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 -I include -I spherehand_amd/csrc "$@" \
  -o tools/$so tools/exp_$which.hip
echo tools/$so

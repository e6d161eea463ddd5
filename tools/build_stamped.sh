#!/bin/bash
# Developer build of libmfhip with -DMF_STAMPS (per-block phase time stamps in gemm_conv_kernel): reflecting-reality_amd/lib/libmfhip_stamps.so
set -eu
root="$(cd "$(dirname "$0")/.." && pwd)"
src="$root/reflecting-reality_amd/csrc"; out="$root/reflecting-reality_amd/lib"
mkdir -p "$out/obj_stamps"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$root/include -DMF_STAMPS=1"
/opt/rocm/bin/hipcc $flags -c "$src/gemm_conv.hip" -o "$out/obj_stamps/gemm_conv.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libmfhip_stamps.so" "$out/obj_stamps/gemm_conv.o" \
    "$out/obj/norm.o" "$out/obj/attention.o" "$out/obj/elementwise.o" "$out/obj/train.o" "$out/obj/frontend.o" "$out/obj/fp8.o"
echo "$out/libmfhip_stamps.so"

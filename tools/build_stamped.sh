#!/bin/bash
# Developer build of libmfhip with -DMF_STAMPS (per-block phase time stamps in gemm_conv_kernel): reflecting-reality_amd/lib/libmfhip_stamps.so
set -eu
root="$(cd "$(dirname "$0")/.." && pwd)"
python "$root/reflecting-reality_amd/_build.py" --variant stamps -D MF_STAMPS=1

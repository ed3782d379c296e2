#!/bin/bash
# Build a variant of the experiments library with extra compile flags: tools/build_variant.sh <name> [-DFLAG=V ...]
# -> infinite-video_amd/libinfv_ltm_v_<name>.so (git-ignored; select it with INFV_LTM_LIBRARY=<path>).  Same-box A/B: tools/ab_libs.sh.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
src="ltm_kernels ltm_chain ltm_chain_batch ltm_uc ltm_dense ltm_psi ltm_capi vqf_kernels split_gemm vqf_capi"
files=""; for s in $src; do files="$files infinite-video_amd/csrc/$s.hip"; done
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -I include -DINFV_EXPERIMENTS "$@" -o infinite-video_amd/libinfv_ltm_v_$name.so $files
echo "built infinite-video_amd/libinfv_ltm_v_$name.so"

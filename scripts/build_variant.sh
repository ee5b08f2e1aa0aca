#!/bin/bash
# Build an alternative liblp_hip with a different conv_igemm source (A/B kernel experiments; select with LP_LIB_OVERRIDE).
# usage: scripts/build_variant.sh <conv_igemm_source.hip> <out.so> [extra hipcc flags]
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/latent_pose_reenactment_amd/csrc; B=$R/latent_pose_reenactment_amd/build
SRC=$1; OUT=$2; shift 2
cp $SRC $C/_variant_conv.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I $R/include -I $C "$@" -c $C/_variant_conv.hip -o /tmp/_variant_conv.o
rm -f $C/_variant_conv.hip
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $B/lp_api.o $B/elementwise.o $B/spectral_norm.o /tmp/_variant_conv.o $B/conv_wgrad.o $B/conv_thin.o

#!/bin/bash
# A/B builds of the library for two questions of VERDICT r4 (run on the CPU container; the variant .so files travel to the GPU box, their objects do not):
#   scripts/libhipadj_Tacqrel.so   -DHIPADJ_TREE_ACQREL=1     the composition tree's ticket as an agent-scope acquire-release RMW (cost per pass?)
#   scripts/libhipadj_Tnz1.so      -DHIPADJ_QUAD_GAUSS_NZ=1   the one-component Gauss instantiation of the quad Tsit5 sweep that came back wrong on the device (still wrong?)
set -e
cd "$(dirname "$0")/../.."
HIPADJ_BUILD_LIB=$PWD/scripts/libhipadj_Tacqrel.so HIPADJ_BUILD_EXTRA="-DHIPADJ_TREE_ACQREL=1" python -m scimlsensitivity_jl_amd.build
HIPADJ_BUILD_LIB=$PWD/scripts/libhipadj_Tnz1.so HIPADJ_BUILD_EXTRA="-DHIPADJ_QUAD_GAUSS_NZ=1" python -m scimlsensitivity_jl_amd.build

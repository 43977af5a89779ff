#!/bin/bash
# Builds drmnet_amd/csrc/libdrmnet_hip_exp.so: the product objects + conv_split2.hip compiled with -DDRM_S2_EXP
# (main-loop experiment switches on DRM_DBG).  Select it with DRM_LIB=exp (see drmnet_amd/_lib.py).
set -e
cd "$(dirname "$0")/.."
python -m drmnet_amd.build
O=drmnet_amd/csrc/_obj
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=off -DDRM_S2_EXP -c drmnet_amd/csrc/conv_split2.hip -o $O/conv_split2_exp.o
objs=$(ls $O/*.o | grep -v conv_split2)
hipcc -shared -fPIC --offload-arch=gfx950 -o drmnet_amd/csrc/libdrmnet_hip_exp.so $objs $O/conv_split2_exp.o
echo built drmnet_amd/csrc/libdrmnet_hip_exp.so

cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_nets.py tests/test_gpu_attn_flash.py -x -q 2>&1 | tail -3
export LAYER_SHAPES="32,896,384,32,64;32,1024,512,16,32;32,640,256,64,128;32,256,128,128,256"
for v in prev - pf3; do
  if [ "$v" = "-" ]; then unset DRM_LIB_PATH; else export DRM_LIB_PATH=$GRAFT_REPO_ROOT/drmnet_amd/csrc/_ab/libdrmnet_hip_$v.so; fi
  echo "[$v]"; python tools/layer_probe.py f16mx 2>&1 | grep -o "resblock.*@[0-9x]*\|conv1x1.*"  | paste - -
done
unset DRM_LIB_PATH
AB_LINES=3 tools/ab_bench.sh drmnet_amd/csrc/_ab/libdrmnet_hip_prev.so - drmnet_amd/csrc/_ab/libdrmnet_hip_pf3.so

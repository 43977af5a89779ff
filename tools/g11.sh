cd $GRAFT_REPO_ROOT
DRM_LIB_PATH=$GRAFT_REPO_ROOT/drmnet_amd/csrc/_ab/libdrmnet_hip_tih.so python -m pytest tests/test_gpu_f16mx.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
for lib in - drmnet_amd/csrc/_ab/libdrmnet_hip_tih.so; do
  if [ "$lib" = "-" ]; then unset DRM_LIB_PATH; else export DRM_LIB_PATH="$GRAFT_REPO_ROOT/$lib"; fi
  echo "[$lib]"; python tools/layer_probe.py f16mx 2>&1 | grep resblock
done; done
AB_LINES=1 tools/ab_bench.sh - drmnet_amd/csrc/_ab/libdrmnet_hip_tih.so

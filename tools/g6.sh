cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in - drmnet_amd/csrc/_ab/libdrmnet_hip_nostage1.so drmnet_amd/csrc/_ab/libdrmnet_hip_nostage2.so; do
  if [ "$lib" = "-" ]; then unset DRM_LIB_PATH; else export DRM_LIB_PATH="$GRAFT_REPO_ROOT/$lib"; fi
  echo "[$lib]"; python tools/layer_probe.py f16mx 2>&1 | grep resblock
done; done

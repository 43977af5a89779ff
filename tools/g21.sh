cd $GRAFT_REPO_ROOT
export LAYER_SHAPES="32,896,384,32,64;32,1024,512,16,32;32,640,256,64,128;32,256,128,128,256"
for v in prev - s1e1 s1e2 s1e3 s1e4 s1e8; do
  if [ "$v" = "-" ]; then unset DRM_LIB_PATH; else export DRM_LIB_PATH=$GRAFT_REPO_ROOT/drmnet_amd/csrc/_ab/libdrmnet_hip_$v.so; fi
  echo "[$v]"; python tools/layer_probe.py f16mx 2>&1 | grep -o "resblock.*@[0-9x]*\|conv1x1.*"  | paste - -
done

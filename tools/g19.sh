cd $GRAFT_REPO_ROOT
export DRM_S2_STAMP_TILE0=1
LAYER_SHAPES="32,896,384,32,64;32,1024,512,16,32;32,640,256,64,128" tools/stamp_probe.sh gpurun_out/stamps_1x1.txt f16mx

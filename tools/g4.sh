cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g4
LAYER_SHAPES="32,256,128,128,256;32,640,256,64,128" tools/stamp_probe.sh gpurun_out/g4/stamps.txt f16mx
grep -A9 "^launch taps 1" gpurun_out/g4/stamps.txt | head -60 > gpurun_out/g4/stamps_1x1.txt

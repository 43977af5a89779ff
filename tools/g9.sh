cd $GRAFT_REPO_ROOT
for lib in drmnet_amd/csrc/_ab/libdrmnet_hip_prev.so -; do
if [ "$lib" = "-" ]; then unset DRM_LIB_PATH; else export DRM_LIB_PATH="$GRAFT_REPO_ROOT/$lib"; fi
echo "[$lib]"; python tools/attn_bench.py 32 16 32 f16x3 2>&1 | tail -1
python tools/attn_bench.py 32 32 32 f16x3 2>&1 | tail -1
done

cd $GRAFT_REPO_ROOT
AB_LINES=1 tools/ab_bench.sh - drmnet_amd/csrc/_ab/libdrmnet_hip_ntstore.so

export DRM_LIB=exp
for d in 0 32 33 34 35 40 43 59 36 48; do DRM_DBG=$d python tools/layer_probe.py 2>&1 | grep resblock; done
echo "--- NODB"
for d in 0 32 33 34 35 40 43 59; do DRM_S2_NODB=1 DRM_DBG=$d python tools/layer_probe.py 2>&1 | grep resblock; done

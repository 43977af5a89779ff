cd $GRAFT_REPO_ROOT
LAYER_SHAPES="8,128,128,128,256;16,128,128,128,256;32,128,128,128,256;64,128,128,128,256;128,128,128,128,256;16,256,256,64,128;32,256,256,64,128;64,256,256,64,128;128,256,256,64,128" python tools/layer_probe.py f16mx
for b in 16 32 64 128; do
  python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-parity-check --no-secondary --no-live-traffic > /tmp/b.log 2>/dev/null
  echo "[batch $b]"; python tools/bsum.py /tmp/b.log | head -3
done

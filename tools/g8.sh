cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_split.py -m gpu -q --collect-only 2>&1 | grep "::" > /tmp/ids.txt
while read id; do
  timeout 120 python -m pytest "$id" -m gpu -x -q 2>&1 | grep -qE "1 passed" && echo "PASS $id" || echo "FAIL $id"
done < /tmp/ids.txt

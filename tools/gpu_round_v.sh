cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for spec in "capture 4 4" "capture 4 2" "capture_noside 4 4" "capture 3 3" "capture 4 4" "capture_keepalive 4 4" "capture_keepalive 4 2" "capture 4 2"; do
  echo "=== $spec"; timeout -k 10 200 python -X faulthandler tools/stream_capture_check.py $spec 2>&1 | grep -v amdgpu | grep "replay\|differ\|per model\|OK\|Error" | cut -c1-200
done

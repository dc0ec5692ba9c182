export JD_VERBOSE=1
for v in "8001 8 7 3" "8001 4 7 3" "8001 4 7 1" "8001 4 2 1" "7003 4 3 1"; do
  echo "== $v"; timeout 50 python tools/rt_pipe_diag.py $v 2>&1 | grep -v "amdgpu.ids" | tail -12 | cut -c1-300
done

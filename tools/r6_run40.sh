export JD_VERBOSE=1
echo "== no plain decode first, three batches, chunk 50"; NOPLAIN=1 NB=3 timeout 60 python tools/rt_pipe_diag.py 8001 4 7 3 50 2>&1 | grep -v amdgpu.ids | tail -14 | cut -c1-300
echo "== no plain decode first, one batch"; NOPLAIN=1 timeout 60 python tools/rt_pipe_diag.py 8001 4 7 3 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300

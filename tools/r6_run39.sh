export JD_VERBOSE=1
echo "== chunk 50, one batch"; timeout 50 python tools/rt_pipe_diag.py 8001 4 7 3 50 2>&1 | grep "pipeline\|step" | cut -c1-300
echo "== three batches, default chunk"; NB=3 timeout 60 python tools/rt_pipe_diag.py 8001 4 7 3 2>&1 | grep "pipeline\|step" | cut -c1-300
echo "== three batches, chunk 50"; NB=3 timeout 60 python tools/rt_pipe_diag.py 8001 4 7 3 50 2>&1 | grep "pipeline\|step" | cut -c1-300

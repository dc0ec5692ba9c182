mkdir -p gpurun_out/r6ab
python -m pytest tests/test_gpu_parity.py -x -q -k "nobody or own_state" > gpurun_out/r6ab/pytest_new.log 2>&1; echo "new rc=$?"; tail -5 gpurun_out/r6ab/pytest_new.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
JD_VERBOSE=1 python tools/run_leg.py clg 1 2>&1 | grep -i "recombine\|state numbers\|per-state words" | head -5
JD_VERBOSE=1 python tools/run_leg.py c3 1 2>&1 | grep -i "recombine\|state numbers\|per-state words" | head -5

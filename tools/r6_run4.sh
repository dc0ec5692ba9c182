mkdir -p gpurun_out/r6d
for v in item_always item_cond; do cp build_ab/lib_$v.so juicer_amd/libjuicer_amd.so; echo "== $v"; python tools/counters_diag.py 2>&1 | grep -v amdgpu.ids; done
JD_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --steps 3 --warmup 1 --utts-per-gpu 4 --arcs 60000 --no-cpu-baseline --no-extra-legs --pipeline-slots 8 > gpurun_out/r6d/b8.out 2> gpurun_out/r6d/b8.err; echo "8 ranks rc=$?"
grep -v "Gloo\|amdgpu.ids" gpurun_out/r6d/b8.err | grep -i "bench.py\|error\|failed" | head -20
tail -c 600 gpurun_out/r6d/b8.out

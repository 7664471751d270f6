cd $GRAFT_REPO_ROOT
python bench.py --workload goku_step --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
run() { python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],4), round(d["kernel_ms"]["lde_forward"],4), round(d["kernel_ms"]["lde_adjoint"],4))'; }
for rep in 1 2 3; do echo "c4 new  $(run c4)"; echo "c4 base $(LDE_LIB_PATH=$PWD/latentdiffeq.jl_amd/liblde_xbase.so run c4)"; done
timeout 600 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_golden.py -q -k "c4 or latentode" 2>&1 | tail -2

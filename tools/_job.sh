cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_autograd_gpu.py -x -q 2>&1 | tail -6
for p in bf16; do timeout 600 python tools/train_bench.py --rows 16384 --precision $p 2>&1 | tail -1; done

cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q -k training_backward 2>&1 | tail -5

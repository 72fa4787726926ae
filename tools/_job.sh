cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 1500 python tools/train_fuzz.py --seeds 0 5000 --budget-s 600 2>&1 | tail -2 | tee gpurun_out/r3h/train_fuzz.json

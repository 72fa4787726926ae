cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 1500 python tools/forward_fuzz.py --seeds 100 5000 --budget-s 1000 2>&1 | tail -3 | tee gpurun_out/r3h/forward_fuzz.json

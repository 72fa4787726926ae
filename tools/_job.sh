set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r3h/gpu_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3h/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > gpurun_out/r3h/smoke.log 2>&1
tail -3 gpurun_out/r3h/gpu_tests.log; tail -1 gpurun_out/r3h/smoke.log
bash tools/profile_round.sh r3h

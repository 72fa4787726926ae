cd $GRAFT_REPO_ROOT
export NORMAL=1 BURST=10 ROUNDS=5 ONLY=p4d,p4dpf
for aux in 2 1 3; do
echo "aux=$aux"
EPI=8 timeout 300 tools/gemm_bench_lean_abl$aux 169283 768 768 118979 2048 2048 77450 4096 4096 2>&1 | cut -c60-260
done

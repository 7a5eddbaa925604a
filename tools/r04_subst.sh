set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not config5" > gpurun_out/pytest_subst.log 2>&1 || { tail -40 gpurun_out/pytest_subst.log; exit 1; }
tail -3 gpurun_out/pytest_subst.log
pk=deepstructuredmixtures_amd
WHAT="d4 c23 h" ROUNDS=2 tools/ab_libs.sh split:$pk/libdsmgp_hip_split.so subst:$pk/libdsmgp_hip_subst.so

set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not config5" > gpurun_out/pytest_reg.log 2>&1 || { tail -40 gpurun_out/pytest_reg.log; exit 1; }
tail -3 gpurun_out/pytest_reg.log
pk=deepstructuredmixtures_amd
WHAT="d4 c23" ROUNDS=3 tools/ab_libs.sh regu:$pk/libdsmgp_hip_regu.so k0:$pk/libdsmgp_hip_k0.so

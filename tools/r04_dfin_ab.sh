set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "diagonal_block_inside or fused_steps_agree or first_bad_minor or single_gp_vs or refit or config1 or prefix" > gpurun_out/pytest_dfin.log 2>&1 || { tail -30 gpurun_out/pytest_dfin.log; exit 1; }
tail -3 gpurun_out/pytest_dfin.log
pk=deepstructuredmixtures_amd
WHAT="h s8 c23" tools/ab_libs.sh prev:$pk/libdsmgp_hip_prev.so back1:$pk/libdsmgp_hip_back1.so back0:$pk/libdsmgp_hip_back0.so front:$pk/libdsmgp_hip_front.so

set -e
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_launch_steps or fused_steps_agree or first_bad_minor or single_gp or config1 or config4 or gradient or finetune or depth4" > gpurun_out/pytest_one.log 2>&1 || { tail -40 gpurun_out/pytest_one.log; exit 1; }
tail -3 gpurun_out/pytest_one.log
pk=deepstructuredmixtures_amd
WHAT="h s8 c23 d4" ROUNDS=2 tools/ab_libs.sh subst:$pk/libdsmgp_hip_subst.so one:$pk/libdsmgp_hip_one.so

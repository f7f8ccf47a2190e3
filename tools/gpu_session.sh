#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 6, session 6: ABI 10 (mtt_ctr_weights, mtt_detloss_*) kernel cases, the new model-level tests (per-parameter gradient bounds,
# dd_label_map_size, imported checkpoint on the device, detection losses), then the rest of the quick GPU tests.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "ctr_weights or detloss" > $O/r06_pytest_f_new_ops.log 2>&1; echo "new ops rc $?"; tail -3 $O/r06_pytest_f_new_ops.log
timeout 900 python -m pytest tests/test_det_losses.py tests/test_gpu_model.py tests/test_gpu_train.py -x -q -m gpu -k "not trajectory and not swin" > $O/r06_pytest_f_model.log 2>&1; echo "model rc $?"; tail -5 $O/r06_pytest_f_model.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "ns6_training_step" > $O/r06_pytest_f_fullsize.log 2>&1; echo "fullsize rc $?"; tail -5 $O/r06_pytest_f_fullsize.log

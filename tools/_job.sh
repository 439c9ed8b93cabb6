#!/bin/bash
python -m pytest tests/test_gpu_window_hot.py tests/test_gpu_window.py tests/test_gpu_wunit.py tests/test_gpu_auto_step.py -x -q 2>&1 | tail -4
SVDF_HOT_ONLY=1 python tools/hot_lane_calibration.py 100000000 0 1024,2048 2>&1 | grep -v "^\[svdf\|amdgpu.ids"

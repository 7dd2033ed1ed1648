#!/bin/bash
mkdir -p gpurun_out/r2y
timeout -k 10 900 python -m pytest tests/test_pipeline.py tests/test_classifier.py tests/test_tools.py -m gpu -q -x -k "pipeline or driver or classif or merger" > gpurun_out/r2y/pytest.log 2>&1
tail -15 gpurun_out/r2y/pytest.log

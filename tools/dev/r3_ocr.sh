#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_ocr_gpu.py -x -q > gpurun_out/r3/ocr_tests.txt 2>&1; echo "rc=$?"; tail -60 gpurun_out/r3/ocr_tests.txt

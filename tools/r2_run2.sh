#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_parity.py -m gpu -q -x -s -k "attention or g3_full or g4_sampler or g2_tiny or c2_full or scale_and" 2>&1 | grep -E "attention act|MPJPE|passed|failed|Error|error|assert" | tail -40 > gpurun_out/r2_tests2.log
cat gpurun_out/r2_tests2.log

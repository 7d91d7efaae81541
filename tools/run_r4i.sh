#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_properties.py -x -q -m gpu -s -k "a6" 2>&1 | grep -E "A6 at 1M|passed|failed|Error|error" | tail -12

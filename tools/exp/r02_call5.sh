#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encoder_overlap" 2>&1 | grep -v "Extension modules" | head -60 | cut -c1-250

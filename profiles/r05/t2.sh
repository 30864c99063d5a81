#!/bin/bash
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r05_t2_tests.log 2>&1; rc=$?; tail -5 gpurun_out/r05_t2_tests.log
exit $rc

#!/bin/bash
# first GPU contact: build check, parity tests
mkdir -p gpurun_out
python -c "import navtex_amd as nv; print(nv.lib.nvx_version(), nv.device_count())" > gpurun_out/first.log 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu >> gpurun_out/first.log 2>&1
echo "exit $?" >> gpurun_out/first.log
tail -40 gpurun_out/first.log

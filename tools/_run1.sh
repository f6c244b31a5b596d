set -e
DCS_CONV_HALO=2 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k conv > gpurun_out/t7.log 2>&1 || { tail -40 gpurun_out/t7.log; exit 1; }
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k conv > gpurun_out/t7b.log 2>&1 || { tail -40 gpurun_out/t7b.log; exit 1; }
DCS_CONV_HALO=0 python tools/conv_bench.py fwd 10 _ > gpurun_out/cb_h0.log 2>&1
DCS_CONV_HALO=2 python tools/conv_bench.py fwd 10 _ > gpurun_out/cb_h2.log 2>&1
DCS_CONV_HALO=0 python tools/conv_bench.py dgrad 10 _ > gpurun_out/cb_h0d.log 2>&1
DCS_CONV_HALO=2 python tools/conv_bench.py dgrad 10 _ > gpurun_out/cb_h2d.log 2>&1

set -e
python tools/conv_bench.py fwd 10 _ > gpurun_out/cb_k32.log 2>&1
DCS_CONV_BK16=1 python tools/conv_bench.py fwd 10 _ > gpurun_out/cb_k16.log 2>&1
python tools/conv_bench.py dgrad 10 _ > gpurun_out/cb_k32d.log 2>&1
DCS_CONV_BK16=1 python tools/conv_bench.py dgrad 10 _ > gpurun_out/cb_k16d.log 2>&1

timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes" 2>&1 | tail -3
bash tools/gpu_ab.sh 2>&1 | head -4
bash tools/gpu_trace.sh r03g era5_1deg_djf30 | grep -v fillBuffer

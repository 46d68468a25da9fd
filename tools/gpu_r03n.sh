timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes" 2>&1 | tail -3
bash tools/gpu_trace.sh r03n era5_1deg_djf30 > /dev/null; grep "seam_driver\|fz_" gpurun_out/r03n/timeline_era5_1deg_djf30.txt
bash tools/gpu_trace.sh r03n era5_025deg_2k > /dev/null; grep "seam_driver\|fz_" gpurun_out/r03n/timeline_era5_025deg_2k.txt
bash tools/gpu_trace.sh r03n era5_025deg_480 > /dev/null; cat gpurun_out/r03n/timeline_era5_025deg_480.txt
for wl in era5_1deg_djf30 era5_025deg_2k; do bash tools/gpu_ab.sh $wl | sed "s/^/$wl /"; done

timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not processes" 2>&1 | tail -3
bash tools/gpu_trace.sh r03m era5_1deg_djf30 > /dev/null; grep -v "^#" gpurun_out/r03m/timeline_era5_1deg_djf30.txt | head -40
bash tools/gpu_trace.sh r03m era5_025deg_2k > /dev/null; head -40 gpurun_out/r03m/timeline_era5_025deg_2k.txt
CTK_ASYNC=0 bash tools/gpu_trace.sh r03m_sync era5_025deg_2k > /dev/null; head -40 gpurun_out/r03m_sync/timeline_era5_025deg_2k.txt
for wl in era5_1deg_djf30 era5_025deg_480 era5_025deg_2k; do bash tools/gpu_ab.sh $wl | sed "s/^/$wl /"; done

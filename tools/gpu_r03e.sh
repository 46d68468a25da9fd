for d in 1 2 3 0; do
export CTK_SD_DBG=$d
echo "== dbg $d"
bash tools/gpu_trace.sh r03e era5_1deg_djf30 2>/dev/null | grep "seam_driver"
bash tools/gpu_trace.sh r03e era5_025deg_2k 2>/dev/null | grep "seam_driver"
done

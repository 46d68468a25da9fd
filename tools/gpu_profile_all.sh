# tools/gpu_profile_all.sh <prefix>: rocprofv3 kernel stats + PMC + bench lines of the four large workloads -> profiles/<prefix>_*
# (<prefix>_bench_<wl>_traced.json: the line of the very process the kernel stats were taken from; <prefix>_bench_<wl>.json: an untraced run)
P=${1:-r03a}
for pair in "era5_1deg_djf30 1deg" "era5_025deg_480 025deg_480" "era5_025deg_10yr 025deg_10yr" "cesm_le_40x30yr cesm_40x30yr"; do
set -- $pair
bash tools/profile.sh ${P}_$2 $1 > /dev/null 2>&1
python tools/summarize_profile.py gpurun_out/${P}_$2 gpurun_out/${P}_$2/sum "$1"
cp gpurun_out/${P}_$2/traced_bench.json gpurun_out/${P}_bench_$2_traced.json; cp gpurun_out/${P}_$2/sum_kernel_stats.csv gpurun_out/${P}_$2_kernel_stats.csv; cp gpurun_out/${P}_$2/sum_pmc.json gpurun_out/${P}_$2_pmc.json; cp gpurun_out/${P}_$2/sum_pmc.md gpurun_out/${P}_$2_pmc.md
rm -rf gpurun_out/${P}_$2
python bench.py --steps 20 --warmup 5 --workload $1 > gpurun_out/${P}_bench_$2.json 2> /dev/null
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload era5_025deg_2k > gpurun_out/${P}_bench_025deg_2k.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --workload era5_1deg_90 > gpurun_out/${P}_bench_1deg_90.json 2>/dev/null
CTK_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 > gpurun_out/${P}_bench_rccl_world1.json 2>/dev/null
CTK_SH_FORCE_SPLIT=1 CTK_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${P}_bench_rccl_world1_force_split.json 2>/dev/null
CTK_DIST_BACKEND=shm python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/${P}_bench_shm_2ranks_one_gpu.json 2>/dev/null
ls -la gpurun_out | grep $P

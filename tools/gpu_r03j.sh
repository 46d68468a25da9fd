for v in 0 1 2 3; do
export CTK_L2D_VARIANT=$v
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('variant $v', round(d['ms_per_step'],4), d['config']['n_tracked'], 'label2d', round(d['kernels_ms']['k_label2d'],4))"
done

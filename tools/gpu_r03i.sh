mkdir -p gpurun_out/r03i
cd /tmp && export TMPDIR=/tmp
CTK_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03i/tr -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-parity-check > $GRAFT_REPO_ROOT/gpurun_out/r03i/trace.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/timeline.py gpurun_out/r03i/tr -2 | cut -c1-110; rm -rf gpurun_out/r03i/tr
CTK_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 --no-parity-check 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('sharded world1', d['ms_per_step'], d['config']['collectives_per_step'])"

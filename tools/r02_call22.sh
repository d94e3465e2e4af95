set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02ap
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02ap/smoke.log 2>&1 || { tail -20 gpurun_out/r02ap/smoke.log; exit 1; }
tail -2 gpurun_out/r02ap/smoke.log
python bench.py --gpus 1 --steps 2 --warmup 1 > gpurun_out/r02ap/bench_driver_like.json 2> gpurun_out/r02ap/bench_driver_like.err
python -c "import json;j=json.load(open('gpurun_out/r02ap/bench_driver_like.json'));print(j['metric'], j['value'], j['n_gpus'], j['steps'], j['warmup'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['traffic'], j['cpu_baseline']['value'], j['b64']['value'])"

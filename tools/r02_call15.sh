set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02aa
timeout -k 10 500 python tools/overlap_probe.py > gpurun_out/r02aa/overlap_probe.txt 2>&1 || { tail -30 gpurun_out/r02aa/overlap_probe.txt; exit 1; }
cat gpurun_out/r02aa/overlap_probe.txt

set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02z
O=gpurun_out/r02z
python tools/ragged_bench.py --batch 64 --utterances 256 > $O/ragged_b64.txt 2>&1
python tools/ragged_bench.py --batch 64 --utterances 1024 >> $O/ragged_b64.txt 2>&1
python tools/ragged_bench.py --batch 128 --utterances 1024 >> $O/ragged_b64.txt 2>&1
cat $O/ragged_b64.txt

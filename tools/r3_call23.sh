# the CLI tests (video frames read back from the PNGs of the hand-written encoder) with the last GPU seconds of the round
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 75 python -m pytest tests/test_gpu_e2e.py -m gpu -q -p no:cacheprovider -k "cli" 2>&1 | tail -2

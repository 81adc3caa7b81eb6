# GIMM-VFI-F CLI test + smoke() at HEAD (token chains, PNG encoder)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 40 python -m pytest tests/test_zz_cli_f.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1
timeout 45 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

for i in 1 2 3; do timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2; done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "${TESTS:-netvlad}" 2>&1 | tail -${LINES:-25}

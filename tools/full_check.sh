cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/final && \
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread > gpurun_out/final/pytest.log 2>&1 ; echo "pytest rc=$?" ; tail -3 gpurun_out/final/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1 ; echo "smoke rc=$?"; tail -2 gpurun_out/final/smoke.log

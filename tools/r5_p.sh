cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py tests/test_product_kernels_gpu.py -q -x 2>&1 | tail -4

# the full GPU suite (as the driver runs it) + the per-case errors of test_engine_vs_oracle_seeded -> gpurun_out/<tag>/
set -u
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-r06h}; mkdir -p gpurun_out/$tag
NK_REQUIRE_FULL=1 timeout 2400 python3 -m pytest tests/ -x -q -m gpu > gpurun_out/$tag/pytest_gpu.txt 2>&1
tail -5 gpurun_out/$tag/pytest_gpu.txt
timeout 900 python3 -m pytest tests/test_engine_gpu.py -q -m gpu -s -k engine_vs_oracle_seeded 2>&1 | grep "value .* gradient .* metric" > gpurun_out/$tag/seeded_errors.txt
sort -t' ' -k1,1 gpurun_out/$tag/seeded_errors.txt | grep float32 | awk '{print $NF}' | sort -g | tail -3

# usage: bash tools/pmc_fetch.sh <script.py> [args]   -- FETCH_SIZE / WRITE_SIZE per kernel (separate passes)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 "$@" > gpurun_out/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 "$@" > gpurun_out/pmc_w.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w

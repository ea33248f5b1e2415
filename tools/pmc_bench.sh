# usage: bash tools/pmc_bench.sh <tag>  -- FETCH_SIZE / WRITE_SIZE passes of the default bench command
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
tag=$1
rm -rf gpurun_out/pmc_${tag}_f gpurun_out/pmc_${tag}_w
timeout 800 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_${tag}_f.log 2>&1
timeout 800 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_${tag}_w.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_${tag}_f gpurun_out/pmc_${tag}_w "1024x1024x1024:f32" "${NK_COMMIT:-?}" > gpurun_out/${tag}_pmc_traffic.json
head -c 1500 gpurun_out/${tag}_pmc_traffic.json

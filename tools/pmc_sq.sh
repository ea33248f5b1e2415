# usage: bash tools/pmc_sq.sh <script.py> [args] -- SQ issue/wait counters per transform kernel (two passes)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/pmc_sq1 gpurun_out/pmc_sq2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq1 -- python3 "$@" > gpurun_out/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq2 -- python3 "$@" > gpurun_out/pmc_sq2.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2

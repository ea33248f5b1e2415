cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_sq1 -- python3 tools/gpu_hartley_only.py > gpurun_out/pmc_sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmc_sq2 -- python3 tools/gpu_hartley_only.py > gpurun_out/pmc_sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq3 -- python3 tools/gpu_hartley_only.py > gpurun_out/pmc_sq3.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 gpurun_out/pmc_sq3
tail -3 gpurun_out/pmc_sq1.log gpurun_out/pmc_sq2.log gpurun_out/pmc_sq3.log

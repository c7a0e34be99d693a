set -u
out=gpurun_out/r2_prof_gemm
mkdir -p $out
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/gemm_profile.py > $out/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- python3 tools/gemm_profile.py > $out/pmc_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $out/pmc_fetch -- python3 tools/gemm_profile.py > $out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $out/pmc_write -- python3 tools/gemm_profile.py > $out/pmc_write.log 2>&1 || exit 1
python3 tools/summarize_profile.py $out $out/summary.md > /dev/null
cat $out/summary.md

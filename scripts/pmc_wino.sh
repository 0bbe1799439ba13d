# usage (GPU box): bash scripts/pmc_wino.sh <tag> -- PMC pass (matrix-pipe utilisation, wave-cycle split, LDS conflicts; own pass, kernel trace only)
# of the direct-vs-Winograd layer microbenchmark -> profiles/<tag>_pmc_mfma.md
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; S=/tmp/dpig_pmc_$TAG; rm -rf $S; mkdir -p $S
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $S/mfma -- python $R/scripts/bench_conv_wino.py > $S/mfma.log 2>&1
cd $R
python scripts/pmc_mfma_summary.py "$(find $S/mfma -name '*.db' | head -1)" $TAG
mkdir -p gpurun_out/profiles_out; cp profiles/${TAG}_pmc_mfma.md gpurun_out/profiles_out/; cat profiles/${TAG}_pmc_mfma.md | tail -8 | cut -c1-200

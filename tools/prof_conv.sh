#!/bin/bash
# PMC passes of one forward conv launch: tools/prof_conv.sh <tag> N Cin H W Cout [f16x3|x6|...]  -> gpurun_out/prof_conv_<tag>/summary.txt
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_conv_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
S=$GRAFT_REPO_ROOT/tools/conv_one.py
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/sq -- python3 $S "$@" > $out/sq.log 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm -- python3 $S "$@" > $out/grbm.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/sq2 -- python3 $S "$@" > $out/sq2.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM --output-format csv -d $out/sq3 -- python3 $S "$@" > $out/sq3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/mfma_util.py conv3x3_split_mfma $out/sq $out/grbm "conv $*" > $out/summary.txt 2>&1
python3 tools/pmc_summary.py conv3x3_split_mfma $out/sq2 >> $out/summary.txt 2>&1
python3 tools/pmc_summary.py conv3x3_split_mfma $out/sq3 >> $out/summary.txt 2>&1
find $out -name "*.csv" -size +2M -delete
cat $out/summary.txt

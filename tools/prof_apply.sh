#!/bin/bash
# Developer script: MFMA-pipe utilisation, wait shares and the shader clock of the headline apply launch, for one setting of the
# environment (e.g. SSTEM_GRAY16=0 tools/prof_apply.sh tag).  Two rocprofv3 --pmc passes (SQ counters / GRBM_GUI_ACTIVE), counters only.
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_apply_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-extra --no-metric-as-worded --no-cpu-baseline --no-live-traffic --steps 30 --prewarm-s 0.3"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/sq -- $B > $out/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm -- $B > $out/grbm.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/sq2 -- $B > $out/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/mfma_util.py sepconv_gray $out/sq $out/grbm "$tag" > $out/summary.txt 2>&1
python3 tools/pmc_summary.py sepconv_gray $out/sq2 >> $out/summary.txt 2>&1
cat $out/summary.txt

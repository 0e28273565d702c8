#!/bin/bash
# Per-kernel PMC evidence for ONE conv geometry through the C ABI (tools/conv_bench.py), one conv pass per profiler run so that
# forward and data-gradient (same kernel) can be told apart.  Counters only (--pmc is never combined with trace domains), the
# program directly after `--`.  FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), SQ counters in a third.
#   gpurun -- 'bash tools/pmc_conv.sh TAG CI CO D H W N DTYPE [passes] [SPLIT]'      ->  gpurun_out/pmc_TAG/summary.txt
# SPLIT > 0: the split-operand passes (conv over cat((xa, xb)), tools/conv_bench.py --cat SPLIT)
set -o pipefail
TAG=$1; CI=$2; CO=$3; D=$4; H=$5; W=$6; N=$7; DT=$8; PASSES=${9:-"fwd dgrad wgrad"}; SPLIT=${10:-0}
CAT=""; if [ "$SPLIT" != "0" ]; then CAT="--cat $SPLIT"; fi
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd $R
for pass in $PASSES; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${pass}_fetch -o pmc -- python3 tools/conv_bench.py $CAT $CI $CO $D $H $W $N 3 $pass $DT > $O/${pass}_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${pass}_write -o pmc -- python3 tools/conv_bench.py $CAT $CI $CO $D $H $W $N 3 $pass $DT > $O/${pass}_write.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $O/${pass}_sq -o pmc -- python3 tools/conv_bench.py $CAT $CI $CO $D $H $W $N 3 $pass $DT > $O/${pass}_sq.log 2>&1 || exit 1
  echo "pmc $TAG $pass done"
done
python3 tools/pmc_conv_summary.py $O $CI $CO $D $H $W $N $DT "$PASSES"

#!/bin/bash
# Which hardware class counter (SQ_INSTS_VALU_ADD_F32 ... _CVT) counts which opcode: the single-opcode streams of tools/valu_calib under the class counters, two passes.
# usage: tools/valu_class_pmc.sh <out dir>   (then: python tools/valu_class_pmc.py <out dir> > profiles/valu_classes.json)
OUT=${1:-/root/repo/gpurun_out/valu_classes}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/cls1 -- /root/repo/tools/valu_calib > $OUT/cls1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/cls2 -- /root/repo/tools/valu_calib > $OUT/cls2.log 2>&1
find $OUT -name "*counter_collection.csv" | head

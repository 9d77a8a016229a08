#!/bin/bash
mkdir -p gpurun_out
{
NGP_FIELD32_FWD=split NGP_FIELD32_BWD=3 python tools/probe_split_accuracy.py 2>&1 | tail -1
NGP_FIELD32_FWD=mfma32 NGP_FIELD32_BWD=2 python tools/probe_split_accuracy.py 2>&1 | tail -1
} | tee gpurun_out/r3ai_accuracy.txt

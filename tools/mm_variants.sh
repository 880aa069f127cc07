#!/bin/bash
# per-launch times of the mbmap launches under phase-skip masks
O=gpurun_out/$1
mkdir -p $O
for m in 0 1 2 3 4 6 7; do
  BN_MM_DBG=$m python tools/kernel_table.py --batch 32 2>/dev/null | grep -E "mbconv:Conv_(114|144|158|188|202|247)" | awk -v m=$m '{print "dbg="m, $1, $NF}' >> $O/mm_dbg.txt
done
for n in 1 2 3 4 6; do
  BN_MBMAP2_NCH=$n python tools/kernel_table.py --batch 32 2>/dev/null | grep -E "mbconv:Conv_(114|144|158|188|202|247)" | awk -v m=$n '{print "nch="m, $1, $NF}' >> $O/mm_nch.txt
  BN_MBMAP2_NCH=$n python tools/kernel_table.py --batch 128 2>/dev/null | grep -E "mbconv:Conv_(114|144|158|188|202|247)" | awk -v m=$n '{print "nch="m" b128", $1, $NF}' >> $O/mm_nch.txt
done
cat $O/mm_dbg.txt $O/mm_nch.txt

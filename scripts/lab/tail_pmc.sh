#!/bin/bash
# lab (round 6): counters of the fused res4 tail (40 images = 1000 tiles) for the product and the shortcut-path variants of DESIGN 9.1b.
# Three --pmc passes (TCP wave latency; TCP -> L2 request latency and pending stalls; SQ / TA vector-memory issue counters), kernel-trace only - never combined with other trace domains.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/tail_pmc; mkdir -p $O
for v in product exp_nores exp_hitres exp_l2res exp_nowait; do
  PE_REPS=8 timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE -d $O/a_$v -o run --output-format csv -- python scripts/lab/tail_quant_probe.py $v 40 > $O/a_$v.log 2>&1
  PE_REPS=8 timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d $O/c_$v -o run --output-format csv -- python scripts/lab/tail_quant_probe.py $v 40 > $O/c_$v.log 2>&1
  [ -n "$ONLY_TCP" ] && continue
  PE_REPS=8 timeout 200 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE -d $O/b_$v -o run --output-format csv -- python scripts/lab/tail_quant_probe.py $v 40 > $O/b_$v.log 2>&1
done
ls $O

#!/bin/bash
# Round profile collection on the GPU box (run through gpurun): rocprofv3 kernel-trace stats of the bench in both
# schedules, three PMC passes (SQ/GRBM, FETCH_SIZE, WRITE_SIZE - never combined with other trace domains), the bench line
# itself.  Everything lands under gpurun_out/$1/; scripts/postprocess_profiles.py turns it into the files under profiles/.
set -u
R=${1:-r02}
OUT=gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-micro"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_serial -o run --output-format csv -- $B --serial-detectors > $OUT/stats_serial.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_two_streams -o run --output-format csv -- $B > $OUT/stats_two_streams.log 2>&1
P="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-micro --serial-detectors"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o run --output-format csv -- $P > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o run --output-format csv -- $P > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o run --output-format csv -- $P > $OUT/pmc_write.log 2>&1
timeout 600 python bench.py --layers $OUT/conv_layers.txt > $OUT/bench.log 2>&1
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
ls $OUT

# ROIAlign: independent 16-byte loads in flight per lane (ROI_MLP; the library in the snapshot was built with the value under test)
mkdir -p gpurun_out/r04_roi_mlp
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r04_roi_mlp/s -o run --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-micro --no-power --serial-detectors > gpurun_out/r04_roi_mlp/log.txt 2>&1
grep -h "roi_align_kernel" gpurun_out/r04_roi_mlp/s/run_kernel_stats.csv | cut -c1-130

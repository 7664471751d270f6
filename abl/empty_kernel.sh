#!/bin/bash
# the duration rocprofv3 reports for kernels that do nothing (run on the GPU box from the repo root)
cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 abl/empty_kernel.hip -o gpurun_out/empty_kernel || exit 1
gpurun_out/empty_kernel > gpurun_out/empty_kernel_plain.json
rm -rf gpurun_out/prof_empty
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_empty -- gpurun_out/empty_kernel > gpurun_out/empty_kernel_prof.json 2> gpurun_out/empty_kernel_prof.err
cat gpurun_out/empty_kernel_plain.json gpurun_out/empty_kernel_prof.json
find gpurun_out/prof_empty -name '*kernel_stats.csv' | head -1 | xargs cat

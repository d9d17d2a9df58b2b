#!/bin/bash
mkdir -p tools/_build
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-pass-failed $1 -I poseprobe_amd/csrc -o tools/_build/gemm256_probe tools/gemm256_probe.hip poseprobe_amd/csrc/pp_error.hip 2>&1 | grep -E "error" 
tools/_build/gemm256_probe ${2:-130944} ${3:-0}

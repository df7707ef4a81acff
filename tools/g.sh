#!/bin/bash
# gpurun wrapper body: the scratch directory does not travel to the GPU box
mkdir -p gpurun_out/r3
eval "$@"

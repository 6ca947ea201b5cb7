#!/bin/bash
# register / LDS / scratch usage of the kernels of one source:  bash scratch/kres.sh trace_lists.hip [grep pattern] [-D flags...]
src=$1; pat=${2:-.}; shift; shift
cd /root/repo/envgs_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -munsafe-fp-atomics -fvisibility=hidden -fno-slp-vectorize "$@" --cuda-device-only -c $src -o /tmp/_kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | paste - - - - - | grep -E "$pat"

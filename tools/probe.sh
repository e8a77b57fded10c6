#!/bin/bash
# development probe: the compiler's resource report for the 33-row contact-class kernels alone (tools/probe_contact.hip); extra -D flags as arguments
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only -c -std=c++17 -O3 -ffp-contract=off -fno-fast-math -mllvm -disable-machine-licm -fno-slp-vectorize -Rpass-analysis=kernel-resource-usage -I../../include -Ihost -Idevice "$@" ../../tools/probe_contact.hip -o /tmp/probe.o 2>&1 | grep -E "error|Name|VGPRs:|VGPRs Spill|SGPRs Spill|Scratch|LDS" | sed 's/.*remark: *//; s/\[-R.*//' | paste - - - - - - | grep -v generic

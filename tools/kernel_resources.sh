#!/bin/bash
# Prints per-kernel register / scratch / LDS usage of a .hip translation unit (hipcc -Rpass-analysis).
# usage: tools/kernel_resources.sh bayesian_torch_amd/csrc/btx_contract_bf16.hip
f="$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(line)print line; line=$0; next} {line=line" | "$0} END{print line}' \
 | sed -E 's/Function Name: //; s/_ZN3btx15contract_kernelI/ck</; s/EEvNS_14ContractParamsE/>/; s/ +/ /g'

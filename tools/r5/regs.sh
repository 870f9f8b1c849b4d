#!/bin/bash
# compile csrc/gemm.hip with the resource-usage remarks and print VGPRs / scratch per 256x256 kernel instantiation
cd /root/repo/uniir_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Rpass-analysis=kernel-resource-usage "$@" -c gemm.hip -o build/gemm.o 2>/tmp/gemm_res.txt
grep -E " error" -A5 /tmp/gemm_res.txt | head -30
grep -E "Function Name|VGPRs:|ScratchSize" /tmp/gemm_res.txt | sed 's/.*remark: [^ ]* //' | paste - - - | grep -E "glds" | sed -E 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/Function Name: _Z16gemm_glds_kernelI//; s/Ev9GemmKArgs//' | awk '{print $1, $2, $3, $4, $5, $6, $7, $8}'

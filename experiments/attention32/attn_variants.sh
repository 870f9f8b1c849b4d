#!/bin/bash
# run the attention microbench for every experimental library given (names of uniir_amd/libuniir_exp_<name>.so)
for n in "$@"; do
  echo "== $n"
  UNIIR_HIP_LIB=$PWD/uniir_amd/libuniir_exp_$n.so MB_ONLY=attn MB_ITEMS=${MB_ITEMS:-1024} python tools/microbench.py 2>&1 | grep -E "attn|Error|error" 
done

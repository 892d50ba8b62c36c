#!/bin/bash
# AddressSanitizer + UBSan run of the host-side C++ (the BPE tokenizer) on adversarial text; CPU only (GPU ASan is not
# available on the pool).  KEDS_BPE_VOCAB=<reference checkout>/src/third_party/open_clip/bpe_simple_vocab_16e6.txt.gz
cd "$(dirname "$0")"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer tokenizer_harness.cpp ../../keds_amd/csrc/tokenizer.cpp -lz -o /tmp/keds_tok_asan && /tmp/keds_tok_asan

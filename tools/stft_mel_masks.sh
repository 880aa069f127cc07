for m in 0 128 256 384 4; do BN_STFT_DBG=$m python tools/kernel_table.py --batch 32 2>&1 | grep stft | cut -c1-40 | sed "s/^/mask $m: /"; done

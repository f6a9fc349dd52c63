#!/bin/bash
# Ablation timing of the matrix-core ELBO kernel: builds diagnostic libraries with one phase removed each and prints the
# per-wave cycle shares (tools/elbo_stamps.py).  Build part runs anywhere (hipcc cross-compiles), the timing part needs a GPU:
#   bash tools/ablate.sh build      then, on the GPU box:      bash tools/ablate.sh
if [ "$1" = "build" ]; then
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -DV2VAE_STAMPS -o tools/libstamps_NONE.so volterra_vae_amd/csrc/*.hip
    for A in ANTIDIAG G U HW; do
        hipcc -O3 --offload-arch=gfx950 -shared -fPIC -DV2VAE_STAMPS -DABLATE_$A -o tools/libstamps_ABLATE_$A.so volterra_vae_amd/csrc/*.hip
    done
    exit 0
fi
for A in NONE ABLATE_ANTIDIAG ABLATE_G ABLATE_U ABLATE_HW; do
    echo "== $A"
    NW=8 STAMPLIB=tools/libstamps_$A.so timeout -k 10 200 python -u tools/elbo_stamps.py 2>&1 | grep "total\|share"
done

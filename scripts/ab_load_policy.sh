#!/bin/bash
# Cache policy of the activation loads (LDS-DMA aux bits) of the pair kernel and of the conv_last strip kernel, over frame sizes:
# four libraries {pair plain / nt} x {conv_last plain / nt} in ONE process per geometry (scripts/ab_pair_libs.py).
# Build first (on the build host):  scripts/ab_load_policy.sh build      then on the GPU box:  bash scripts/ab_load_policy.sh > gpurun_out/ab_load_policy.txt
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  KFILE="kernels_pair.hip kernels_last.hip" scripts/ablate_pair.sh pp "-DKP_DMA_AUX=0 -DKL_DMA_AUX=0" np "-DKP_DMA_AUX=2 -DKL_DMA_AUX=0" pn "-DKP_DMA_AUX=0 -DKL_DMA_AUX=2" nn "-DKP_DMA_AUX=2 -DKL_DMA_AUX=2"
  exit 0
fi
echo "libraries: pp = plain loads in both kernels, np = pair kernel nt / conv_last plain, pn = pair plain / conv_last nt, nn = both nt (shipped); 'unfused' = pp's library, one layer per launch"
L="pp=reve_amd/ablp_pp.so np=reve_amd/ablp_np.so pn=reve_amd/ablp_pn.so nn=reve_amd/ablp_nn.so"
for geo in "960 540 60" "1280 720 40" "1600 900 30" "1920 1080 30" "2560 1080 20" "2560 1440 16" "3840 2160 10"; do
  set -- $geo
  echo "== $1x$2 (one fp16 activation: $(( ($1 + 2) * ($2 + 2) * 128 / 1000000 )) MB)"
  W=$1 H=$2 N=$3 ROUNDS=5 timeout 300 python3 scripts/ab_pair_libs.py $L 2>&1 | grep -o "^.*conv_last[^|]*"
done
